"""Register budgets the pipeline's overlap depends on, checked at build time (no GPU: hipcc cross-compiles and reports).

final_select_kernel<8> must fit on a CU NEXT TO a scan block -- 4 waves/SIMD x <= 96 VGPRs beside the scan's 2 x 56 -- or the async
select of the one-query pipeline waits for scan blocks to leave: in round 5 a few lines added to that kernel took it from 90 to 106
VGPRs and the headline from 153 to 204 us per 1 M-row step, with every parity test green."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _usage(src):
    r = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fno-gpu-rdc", "-x", "hip", "--offload-device-only",
                        "-c", os.path.join(ROOT, "semtools_amd", "csrc", src), "-o", "/dev/null", "-Rpass-analysis=kernel-resource-usage"],
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    out, name = {}, None
    for line in r.stderr.splitlines():
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            name = m.group(1)
            out[name] = {}
            continue
        m = re.search(r"remark:\s+([A-Za-z ]+(?:\[[^\]]+\])?):\s+(\d+)", line)
        if m and name:
            out[name][m.group(1).strip()] = int(m.group(2))
    return out


@pytest.mark.timeout(900)
def test_select_kernel_fits_beside_a_scan_block_and_nothing_spills():
    u = _usage("scan_kernels.hip")
    sel = [v for k, v in u.items() if "final_select_kernelILi8ELb0" in k]
    assert len(sel) == 1, list(u)
    assert sel[0]["VGPRs"] <= 96, sel[0]
    scans = {k: v for k, v in u.items() if "scan_topk_kernel" in k}
    assert scans
    for k, v in scans.items():
        assert v["VGPRs Spill"] == 0 and v["VGPRs"] <= 128, (k, v)      # 8 waves per CU: two per SIMD with room for the select
    one_query = [v for k, v in scans.items() if "ILi1ELi4ELb1ELb0" in k]
    assert len(one_query) == 2 and all(v["VGPRs"] <= 64 for v in one_query), one_query   # the headline instantiation (static deal and the
                                                                                          # scan_steal A/B form): beside 4 x 96 of the select
