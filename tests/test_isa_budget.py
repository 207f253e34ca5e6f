"""Static budgets of the headline kernels, checked on the hipcc listing (no GPU): registers, scratch, LDS and the vector-ALU
content of the hottest basic block.  Each number below has a measured consequence: a second wave per
SIMD needs <= 256 registers, a third <= 168; a zero-VALU k-loop was worth 10 % on the dominant kernel; `k_dgrad_pix_z` with 256
registers + scratch runs at HALF speed (profiles/r05_ah_early_pixdb.log).  The test re-derives the table from the current
sources, so a change that costs occupancy or spills shows up before it reaches a GPU box."""
import importlib.util
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")

# mangled-name fragment: (max vgpr + agpr, max scratch instructions, max LDS bytes, max VALU in the hottest block, work-groups
# per CU the launcher counts on)
BUDGET = {
    "k_fwd_glds_ztILi128ELi64ELi2ELi2E": (168, 0, 0, 16, 3),            # conv2 forward (dominant): 3 work-groups per CU, dynamic LDS
    "k_fwd_glds_zILi64ELi64ELi2ELi2E": (128, 0, 32768, 8, 4),           # fc forward of a rollout step
    "k_fwd_glds_zILi128ELi128ELi2ELi2E": (256, 32, 65536, 0, 2),        # fc forward / data gradient (mask prefetch: known spills outside the loop)
    "k_dgrad_quadrow_zILi128ELi128ELi2ELi2E": (256, 0, 66560, 28, 2),   # conv2 data gradient (double-buffered fragments)
    "k_dgrad_pix_zILi128ELi64ELi2ELi2E": (200, 0, 49152, 12, 2),        # conv3 data gradient
    "k_wgrad_glds_zILi128ELi128ELi2ELi2E": (256, 0, 65536, 1, 2),       # fc weight gradient
    "k_wgrad_imgILi32ELi20ELi20ELi4ELi2ELi2E": (256, 0, 145408, 28, 1),  # conv2 weight gradient (persistent, one work-group per CU)
    "k_wgrad_imgILi64ELi9ELi9ELi3ELi1ELi1E": (256, 0, 69632, 0, 2),     # conv3 weight gradient
    "k_fwd_imgILi64ELi9ELi9ELi3ELi1ELi2ELi1ELi7E": (256, 0, 81920, 80, 2),  # conv3 forward (SF_IMG_FLIP: planes padded to 16 chunks, 3 x 24 KiB; two work-groups per CU)
    "k_conv1_u8_bf16_wILb0E": (256, 0, 0, 57, 2),                       # conv1 forward on u8 frames (dynamic LDS)
    "k_conv1_wgrad_bf16ILb0E": (256, 0, 0, 40, 2),                      # conv1 weight gradient
}


@pytest.fixture(scope="module")
def listing(tmp_path_factory):
    if not (os.path.isfile(HIPCC) or shutil.which(HIPCC)):
        pytest.skip("hipcc not available")
    out = tmp_path_factory.mktemp("isa") / "sf_nn.s"
    cmd = [HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-I" + os.path.join(ROOT, "include"),
           "-I" + os.path.join(ROOT, "sample_factory_amd", "csrc"), "-S", "--cuda-device-only",
           os.path.join(ROOT, "sample_factory_amd", "csrc", "sf_nn.hip"), "-o", str(out)]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    spec = importlib.util.spec_from_file_location("isa_stats", os.path.join(ROOT, "tools", "isa_stats.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return {k["name"]: k for k in mod.kernel_stats(out.read_text())}


@pytest.mark.parametrize("fragment", sorted(BUDGET))
def test_headline_kernel_stays_inside_its_budget(listing, fragment):
    regs, scratch, lds, valu, wgs = BUDGET[fragment]
    found = [k for name, k in listing.items() if fragment in name]
    assert len(found) == 1, f"{fragment}: {len(found)} instantiations in the listing"
    k = found[0]
    used = k["vgpr"] + k["agpr"]
    assert used <= regs, f"{fragment}: {used} registers > {regs}"
    assert k["scratch"] <= scratch, f"{fragment}: {k['scratch']} scratch instructions (spills) > {scratch}"
    assert (k["lds"] or 0) <= lds, f"{fragment}: {k['lds']} bytes of static LDS > {lds}"
    assert k["hot"]["valu"] <= valu, f"{fragment}: {k['hot']['valu']} vector-ALU instructions in the hottest block > {valu}"
    assert k["hot"]["mfma"] >= 32, "the hottest block is the MFMA loop"
    # occupancy the dispatch relies on: 4 waves per work-group, one per SIMD -> `wgs` waves per SIMD share 512 registers
    # (allocation granule 8), and `wgs` work-groups share 160 KB of LDS
    granule = (used + 7) // 8 * 8
    assert wgs * granule <= 512, f"{fragment}: {wgs} waves per SIMD need {wgs * granule} registers"
    assert wgs * (k["lds"] or 0) <= 160 * 1024, f"{fragment}: {wgs} work-groups need {wgs * (k['lds'] or 0)} bytes of LDS"


def test_zero_valu_loops_are_zero_valu(listing):
    """the k-loops of the `_z` family carry no vector-ALU instruction at all: hottest block of the two
    instantiations whose hottest block IS the steady-state loop"""
    for fragment in ("k_fwd_glds_zILi128ELi128ELi2ELi2E", "k_wgrad_imgILi64ELi9ELi9ELi3ELi1ELi1E"):
        k = next(v for n, v in listing.items() if fragment in n)
        assert k["hot"]["valu"] == 0 and k["hot"]["mfma"] >= 64, (fragment, k["hot"])
