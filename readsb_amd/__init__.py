"""readsb_amd — MI355X-native Mode-S demodulator hot path (libmodes_gpu.so) and its thin
ctypes binding.  The product is the C-ABI library under readsb_amd/csrc (include/modes_gpu.h);
Python is only used to drive tests and the benchmark."""
from .binding import (  # noqa: F401
    FMT_UC8, FMT_SC16, FMT_SC16Q11, MSG_DTYPE, FIELDS_DTYPE, Demodulator, MgpuError, lib_path, load_library,
)
