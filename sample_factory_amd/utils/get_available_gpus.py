"""GPU enumeration WITHOUT initialising the HIP runtime in the calling process (reference:
sample_factory/utils/get_available_gpus.py:5-44 spawns `python -m ...get_available_gpus` so the parent can still set the
visibility variable afterwards).  On ROCm the KFD topology in sysfs answers the question directly: every node with
`simd_count > 0` is a GPU agent; the subprocess is only the fallback."""
from __future__ import annotations

import glob
import os
import sys


def _kfd_gpu_count() -> int:
    n = 0
    for props in glob.glob("/sys/class/kfd/kfd/topology/nodes/*/properties"):
        try:
            with open(props) as f:
                for line in f:
                    if line.startswith("simd_count"):
                        n += int(line.split()[1]) > 0
                        break
        except OSError:
            continue
    return n


def get_gpus_without_triggering_pytorch_cuda_initialization(envvars=None) -> str:
    n = _kfd_gpu_count()
    if n == 0 and os.path.exists("/dev/kfd"):  # topology unreadable: ask a child process
        import subprocess
        out = subprocess.run([sys.executable, "-m", "sample_factory_amd.utils.get_available_gpus"], capture_output=True,
                             env=dict(envvars if envvars is not None else os.environ))
        return out.stdout.decode().strip()
    return ",".join(str(g) for g in range(n))


def main() -> int:
    import torch
    print(",".join(str(g) for g in range(torch.cuda.device_count())))
    return 0


if __name__ == "__main__":
    sys.exit(main())
