"""The build's spill guard (vist3a_amd/build.py): hipcc's kernel-resource-usage remarks are parsed per source file and a kernel that needs
more than MAX_SCRATCH_BYTES_PER_LANE of scratch fails the build - a slow kernel that still computes the right numbers is invisible to every
parity test (late round 4: two attention instantiations went from 52 / 0 to 3.3 KB of scratch per lane after a one-line change elsewhere in
the template, and a 21 us launch took 950 us)."""
import json

from vist3a_amd import build

REMARKS = """\
a.hip:29:1: remark: Function Name: _ZN1A4kernIfEEvv [-Rpass-analysis=kernel-resource-usage]
a.hip:29:1: remark:     SGPRs: 40 [-Rpass-analysis=kernel-resource-usage]
a.hip:29:1: remark:     VGPRs: 168 [-Rpass-analysis=kernel-resource-usage]
a.hip:29:1: remark:     ScratchSize [bytes/lane]: 3336 [-Rpass-analysis=kernel-resource-usage]
a.hip:29:1: remark:     Occupancy [waves/SIMD]: 3 [-Rpass-analysis=kernel-resource-usage]
a.hip:90:1: remark: Function Name: other_kernel [-Rpass-analysis=kernel-resource-usage]
a.hip:90:1: remark:     VGPRs: 54 [-Rpass-analysis=kernel-resource-usage]
a.hip:90:1: remark:     ScratchSize [bytes/lane]: 0 [-Rpass-analysis=kernel-resource-usage]
a.hip:90:1: remark:     Occupancy [waves/SIMD]: 8 [-Rpass-analysis=kernel-resource-usage]
"""


def test_resource_remarks_are_parsed_per_kernel():
    r = build.kernel_resources(REMARKS)
    assert r == {"_ZN1A4kernIfEEvv": {"vgprs": 168, "scratch": 3336, "occupancy": 3}, "other_kernel": {"vgprs": 54, "scratch": 0, "occupancy": 8}}
    assert build.MAX_SCRATCH_BYTES_PER_LANE <= 512


def test_built_kernels_stay_under_the_scratch_limit():
    files = sorted(build.OBJ.glob("*.resources.json"))
    if not files:
        import pytest
        pytest.skip("library not built in this checkout")
    worst = {}
    for f in files:
        for k, v in json.loads(f.read_text()).items():
            if v["scratch"] > build.MAX_SCRATCH_BYTES_PER_LANE:
                worst[f"{f.stem}:{k}"] = v["scratch"]
    assert not worst, worst
