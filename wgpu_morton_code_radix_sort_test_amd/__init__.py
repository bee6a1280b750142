"""MI355X-native Morton-key radix sort: Python host side over the C ABI of libmrs.so.

The product is the shared library (`csrc/libmrs.so`, declared in `include/mrs.h`);
this package is the thin host mirror used by the tests, the benchmark and the
multi-GPU driver.  torch supplies device memory, streams and torch.distributed.
"""
from ._lib import MrsError, load_library, library_path  # noqa: F401
from .api import (  # noqa: F401
    ALGO_HYBRID,
    ALGO_ONESWEEP,
    ALGO_THREE_KERNEL,
    Context,
    PlanInfo,
    SortPlan,
)
from . import scene  # noqa: F401
from .harness import run_compute_shaders, run_headless  # noqa: F401
