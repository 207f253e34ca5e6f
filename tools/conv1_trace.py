"""Phase breakdown of k_conv1_u8_bf16 (experiment build with -DSF_CONV1_TRACE=1, see tools/experiments/r05_conv1.sh):
per-wave shader-cycle sums of the strip loop's phases, averaged per (wave, strip unit).
    SF_HIP_LIB=$PWD/build/variants/libsf_hip_c1trace.so python tools/conv1_trace.py [n]"""
import ctypes, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sample_factory_amd import lib
from kbench import desc

PHASES = ["wait prefetched bytes (vmcnt 0)", "convert + LDS writes", "barrier 1", "load issue + fragment reads + MFMAs",
          "epilogue issue (stores / staging writes)", "barrier 2", "staged tile -> global (whole-line stores)"]

def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
    d = desc(4, 84, 84, 32, 8, 4, 1)
    dll = ctypes.CDLL(os.environ["SF_HIP_LIB"])
    x = torch.randint(0, 256, (n, 4, 84, 84), dtype=torch.uint8, device="cuda")
    w = torch.randn((256, 32), device="cuda") / 16
    b = torch.zeros(32, device="cuda")
    out = torch.empty((n * 400, 32), device="cuda")
    acc = (ctypes.c_ulonglong * 12)()
    for _ in range(3): lib.conv_fwd(x, 4 * 84 * 84, None, 0, w, b, out, n, d, None)
    assert dll.sf_debug_conv1_trace(acc) == 0
    reps = 5
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): lib.conv_fwd(x, 4 * 84 * 84, None, 0, w, b, out, n, d, None)
    e.record(); torch.cuda.synchronize()
    assert dll.sf_debug_conv1_trace(acc) == 0
    units = acc[7]  # summed over waves
    tot = sum(acc[i] for i in range(7))
    print(f"n={n}: {s.elapsed_time(e) / reps * 1e3:.1f} us per launch (instrumented); {units // reps} wave-units per launch; "
          f"{tot / units:.0f} shader cycles per wave and unit")
    print(f"  strip loop per wave: {acc[8] / acc[10]:.0f} s_memtime ticks in {acc[9] / acc[10] / 100:.1f} us of the 100 MHz clock "
          f"-> {acc[8] / acc[9] / 10:.3f} ticks per ns; {acc[10] // reps} waves per launch")
    for i, name in enumerate(PHASES):
        print(f"  {name:42s} {acc[i] / units:8.0f} cycles  {acc[i] / tot:6.1%}")

if __name__ == "__main__":
    main()
