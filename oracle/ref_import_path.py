"""Where the reference lives for this process — path logic only, NO stub installation (bench.py's own process imports
this to decide whether the cpu_baseline leg can run the reference; the stubs of oracle/ref_import.py are installed only
inside the baseline subprocess).  TEST / BENCH INFRASTRUCTURE."""
import os

_REF_ZIP = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_ref", "sample_factory_ref.zip")
REFERENCE_ROOT = os.environ.get("SF_REFERENCE_ROOT") or (
    "/root/reference" if os.path.isdir("/root/reference/sample_factory") else _REF_ZIP)  # a .zip is imported by zipimport


def reference_available() -> bool:
    return os.path.exists(REFERENCE_ROOT)
