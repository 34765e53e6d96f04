#!/usr/bin/env python3
"""Print name, calls, average us of a rocprofv3 *kernel_stats.csv (names cut at the first parenthesis)."""
import csv
import sys
for r in csv.DictReader(open(sys.argv[1])):
    print(f"{r['Name'].split('(')[0][:50]:50s} {int(r['Calls']):5d} calls  avg {float(r['AverageNs']) / 1e3:8.2f} us  min {float(r['MinNs']) / 1e3:8.2f}  max {float(r['MaxNs']) / 1e3:8.2f}  {float(r['Percentage']):5.1f} %")
