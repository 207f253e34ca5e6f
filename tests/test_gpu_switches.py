"""Every A/B switch of the network kernels (SF_CONV1_BF16, SF_CONV1_IMG, SF_DGRAD_PIX, SF_WGRAD_GLDS, SF_WGRAD_IMG, SF_RELU_MASK, SF_GLDS_CFG,
SF_FWD_IMG, SF_GLDS_SPLITK, SF_GLDS_FC64, SF_GLDS_SMALL64, SF_GLDS_SPLIT64, SF_TAP_PERM, SF_XCD_ROWS, SF_GLDS_ZL, SF_CONV1_WIDE, SF_XCD_RASTER, SF_DGRAD_LPT, SF_LINEAR_NARROW, SF_REDUCE_TREE; recurrent passes: SF_SEQ_BWD_REGW, SF_SEQ_FWD_X — read once per process) selects a different kernel for the same operation; each must pass
the same kernel-vs-torch tests as the default dispatch.  One pytest subprocess per group of independent (different
operation) non-default settings; the two tests that assert which kernel / plan the DEFAULT dispatch picks are left out
where the switch under test changes exactly that."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SELECT = "(vs_torch or fuzz or in_place)"  # the kernel-level numerics tests of tests/test_gpu_nn.py

GROUPS = [
    # the register-staged / im2col generation for every operation
    ("SF_CONV1_BF16=0 SF_CONV1_IMG=0 SF_DGRAD_PIX=0 SF_WGRAD_GLDS=0 SF_WGRAD_IMG=0 SF_FWD_IMG=0", " and not lds_image_forward_conv3"),
    # alternative tilings of the LDS-DMA kernels
    # ... and conv1 on the f32 strip-image kernels instead of the exact-product bf16 ones
    ("SF_CONV1_BF16=0 SF_DGRAD_PIX=2 SF_WGRAD_GLDS=2 SF_WGRAD_IMG=0 SF_GLDS_CFG=2 SF_GLDS_ZL=0 SF_DGRAD_ZL=2 SF_WGRAD_ZL=0", ""),
    ("SF_DGRAD_PIX=3 SF_WGRAD_GLDS=3 SF_WGRAD_IMG=1 SF_RELU_MASK=0 SF_GLDS_SPLITK=0", " and not splitk_small_grids and not relu_sign_bits"),
    # the tiled kernels for the narrow / small linear layers, the serial reduction of partials
    # ... conv1 forward with per-wave dword stores instead of the LDS-staged whole-line ones
    ("SF_LINEAR_NARROW=0 SF_REDUCE_TREE=0 SF_CONV1_WIDE=0", " and not narrow_linear"),
    # round-4 dispatch changes switched off: fc forward of a rollout step back on 128x128 tiles x K slices, plain block
    # order instead of the XCD-aware one, conv3 data-gradient rows in index order instead of longest first
    # (+ round 5: small inference launches back on the register-staged kernels)
    ("SF_GLDS_FC64=0 SF_XCD_RASTER=0 SF_DGRAD_LPT=0 SF_GLDS_SMALL64=0 SF_GLDS_SPLIT64=0 SF_TAP_PERM=1 SF_XCD_ROWS=1", " and not 64x64_unsplit"),
]


@pytest.mark.parametrize("switches,minus", GROUPS)
def test_kernel_numerics_under_non_default_switches(switches, minus):
    env = dict(os.environ, **dict(kv.split("=") for kv in switches.split()))
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_gpu_nn.py"), "-q", "-x", "-m", "gpu",
                        "-k", SELECT + minus, "-p", "no:cacheprovider"], cwd=ROOT, env=env, capture_output=True, text=True,
                       timeout=900)
    tail = r.stdout[-1500:]
    assert r.returncode == 0 and " passed" in tail and "failed" not in tail, f"{switches}:\n{tail}\n{r.stderr[-500:]}"


def test_rnn_sequence_passes_with_lds_weight_backward():
    """SF_SEQ_BWD_REGW=0 SF_SEQ_FWD_X=0 SF_LINEAR_DUAL=0 SF_WGRAD_GLDS_K64=0 SF_DGRAD_LINEAR64=0: the backward sequence
    passes with the W_hh slice in LDS (16 hidden units per work-group; the dispatch for row groups of more than 64 rows), the
    forward passes fed by a separate gx GEMM, the inference step as two projections and the input projection's gradients on
    the general kernels must pass the same fused-sequence tests and the config-5 reference replays as the default kernels"""
    env = dict(os.environ, SF_SEQ_BWD_REGW="0", SF_SEQ_FWD_X="0", SF_LINEAR_DUAL="0", SF_WGRAD_GLDS_K64="0", SF_DGRAD_LINEAR64="0")
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_gpu_rl_kernels.py"),
                        os.path.join(ROOT, "tests", "test_gpu_parity_c2_c5.py"), "-q", "-x", "-m", "gpu",
                        "-k", "fused_lstm_sequence or fused_gru_sequence or config5", "-p", "no:cacheprovider"], cwd=ROOT, env=env,
                       capture_output=True, text=True, timeout=900)
    tail = r.stdout[-1500:]
    assert r.returncode == 0 and " passed" in tail and "failed" not in tail, f"{tail}\n{r.stderr[-500:]}"
