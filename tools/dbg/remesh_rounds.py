import sys; sys.path.insert(0,'.')
from drt_amd import remesh_gpu as RG
RG.DEBUG=True
sys.argv=['x','0.9']
exec(open('tools/ubench/remesh_probe.py').read().replace('for rep in range(3):','for rep in range(1):'))
