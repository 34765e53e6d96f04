"""Code-generation guards (cross-compiled for gfx950 here, no GPU needed).

k_refit (drt_amd/csrc/drt_build.hip) hands child boxes between workgroups WITHOUT fences: the producer writes three 8-byte
granules with relaxed agent-scope atomic stores, drains them, bumps the node's counter; the second arriver reads the
sibling's granules with relaxed agent-scope atomic loads.  On MI355X that is only correct if the stores are write-through
(`sc1`), the loads bypass the non-coherent L1 (`sc1`), and an `s_waitcnt vmcnt(0)` sits between the last granule store and
the counter atomic.  The rebuild-under-load GPU test checks the behaviour; this test pins the instructions, so that a
compiler or source change that silently drops one of them fails HERE."""
import os
import re
import subprocess

import pytest

from conftest import ROOT
from drt_amd import build


@pytest.fixture(scope="module")
def build_isa(tmp_path_factory):
    out = tmp_path_factory.mktemp("isa") / "drt_build.s"
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    subprocess.check_call([hipcc] + build.FLAGS + ["-S", "--cuda-device-only", "-o", str(out), os.path.join(build.CSRC, "drt_build.hip")],
                          stderr=subprocess.DEVNULL)
    return out.read_text()


def _kernel_body(isa, name):
    m = re.search(rf"^(_Z\d+{name}[A-Za-z0-9_]*):.*?$(.*?)^\s*s_endpgm", isa, re.S | re.M)
    assert m, f"kernel {name} not found in the ISA"
    return [ln.split(";")[0].strip() for ln in m.group(2).splitlines() if ln.strip() and not ln.strip().startswith((";", "."))]


def test_refit_handoff_is_write_through_drained_and_l1_bypassing(build_isa):
    body = _kernel_body(build_isa, "k_refit")
    mem = [ln for ln in body if ln.startswith(("global_", "buffer_", "flat_", "s_waitcnt vmcnt"))]
    at = [k for k, ln in enumerate(mem) if ln.startswith("global_atomic_add")]
    assert len(at) == 1, mem                                    # the node counter
    k = at[0]
    assert re.search(r"\bsc0\b", mem[k]), mem[k]                 # returning atomic (old value decides who owns the node)
    # producer side: exactly three 8-byte write-through granule stores, then a full drain, directly before the atomic
    assert mem[k - 1] == "s_waitcnt vmcnt(0)", mem[k - 4:k + 1]
    stores = mem[k - 4:k - 1]
    assert all(s.startswith("global_store_dwordx2") and re.search(r"\bsc1\b", s) for s in stores), stores
    # no fence instructions anywhere in the kernel (the point of this design: 30 us instead of 235 us)
    assert not any(ln.startswith(("buffer_wbl2", "buffer_inv")) for ln in body)
    # consumer side: after the atomic's result is waited for, the sibling granules are read with three L1-bypassing loads
    after = mem[k + 1:]
    assert after[0] == "s_waitcnt vmcnt(0)", after[:3]
    loads = [ln for ln in after if ln.startswith("global_load_dwordx2")]
    assert len(loads) == 3 and all(re.search(r"\bsc1\b", ln) for ln in loads), loads
    # and no OTHER store to node memory precedes the granule stores inside the loop without sc1
    loop_stores = [ln for ln in mem[:k] if ln.startswith("global_store_dwordx2")]
    assert all(re.search(r"\bsc1\b", ln) for ln in loop_stores), loop_stores


def test_traversal_kernels_fit_eight_waves_per_simd():
    """k_trace must stay within 64 VGPRs (8 waves per SIMD hide the dependent node fetches) and its persistent loop must not touch scratch
    memory.  The loop is everything in front of the first barrier (the epilogue -- the second pass of the workgroup that retires last, a few
    rays per launch at most -- starts with one); the seeded instantiation keeps two registers of that epilogue in scratch, which is allowed."""
    out = subprocess.run([os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")] + build.FLAGS + ["-S", "--cuda-device-only", "-o", "-",
                         os.path.join(build.CSRC, "drt_trace.hip")], check=True, capture_output=True, text=True).stdout      # (every k_trace instantiation lives there)
    meta = re.findall(r"\.name:\s+(_Z7k_trace\S+).*?\.private_segment_fixed_size:\s+(\d+).*?\.vgpr_count:\s+(\d+)", out, re.S)
    assert len(meta) >= 3, "k_trace instantiations not found"
    for name, scratch, vgprs in meta:
        assert int(vgprs) <= 64, (name, vgprs)
        # k_trace<false, 0, true> (temporal hit seeds: opt-in, off by default) and k_trace<false, 1> (boundary B1: T and ID per ray number) may keep
        # registers of their EPILOGUE in scratch; the pipeline's instantiations may not use scratch at all
        relaxed = name.startswith(("_Z7k_traceILb0ELi0ELb1E", "_Z7k_traceILb0ELi1ELb0E"))
        assert int(scratch) <= (32 if relaxed else 0), (name, scratch)
    kernels = re.findall(r"^(_Z7k_trace\w+):.*?\n(.*?)s_endpgm", out, re.S | re.M)
    assert len(kernels) >= 3
    for name, text in kernels:
        loop = text.split("s_barrier")[0]
        assert "v_fma_mix_f32" in loop, name                  # (it IS the traversal loop)
        assert "scratch_" not in loop, (name, "scratch traffic inside the persistent loop")


def test_inner_visit_reads_bounds_as_float16_subnormals_and_issues_four_loads():
    """The slab test of k_trace feeds the quantised bounds to v_fma_mix_f32 as float16 SUBNORMALS (byte * 2^-24): that is only
    the fma of drt_traverse.h's host form if the kernel runs with float16 denormals enabled (descriptor mode 3), and it is only
    worth it while no byte->float converts are left.  The child references must be loaded beside the three bound chunks (four
    16-byte loads of one node in a row), not inside the "a child was hit" branch as a second, dependent round trip."""
    out = subprocess.run([os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")] + build.FLAGS + ["-S", "--cuda-device-only", "-o", "-",
                         os.path.join(build.CSRC, "drt_trace.hip")], check=True, capture_output=True, text=True).stdout      # (every k_trace instantiation lives there)
    kernels = re.findall(r"^(_Z7k_trace\w+):.*?\n(.*?)s_endpgm", out, re.S | re.M)
    assert len(kernels) >= 2
    for name, text in kernels:
        body = [ln.split(";")[0].strip() for ln in text.splitlines() if ln.strip() and not ln.strip().startswith((";", "."))]
        body = [ln for ln in body if not ln.startswith("s_nop")]          # (hazard padding between two loads is not a separation)
        assert sum(ln.startswith("v_fma_mix_f32") for ln in body) >= 24, name
        assert sum(ln.startswith("v_perm_b32") for ln in body) >= 12, name
        assert not any(ln.startswith("v_cvt_f32_ubyte") for ln in body), name
        loads = [k for k, ln in enumerate(body) if ln.startswith("global_load_dwordx4")]
        runs = [k for k in loads if all(body[k + j].startswith("global_load_dwordx4") for j in range(4) if k + j < len(body)) and k + 3 < len(body)]
        assert runs, (name, "the four node loads are not issued together")
        desc = re.search(r"\.amdhsa_kernel " + re.escape(name) + r"\b(.*?)\.end_amdhsa_kernel", out, re.S).group(1)
        assert re.search(r"\.amdhsa_float_denorm_mode_16_64\s+3", desc), name
        assert re.search(r"\.amdhsa_float_denorm_mode_32\s+3", desc), name
