"""yacrd_amd — MI355X-native bad-region engine behind yacrd's BadPart boundary.

The product is the C-ABI library `yacrd_amd/lib/libyacrd_hip.so` (HIP kernels for gfx950 +
host launcher, include/yacrd_engine.h) and the C++ host (`yacrd_amd/bin/yacrd`).  This Python
package is a thin ctypes binding used by tests and bench.py; it has no CPU fallback: without
the built library, or without a gfx950 device, it raises.
"""
from .engine import (  # noqa: F401
    Engine, EngineError, NeedsHostParser, Result, lib_path, load_library, build, partition_reads, run_partitioned, ingest_overlaps,
    NOT_BAD, CHIMERIC, NOT_COVERED, TYPE_NAMES, F_FORCE_GENERAL, F_FORCE_LDS_SORT, F_XLANE_DS, F_WAVE_ONLY, F_NO_HALVES, F_TIMING_FULL, F_NO_PREDICTION, F_NO_FUSED_LAUNCH, F_NO_PREFILTER, F_COUNT_PREFILTERED, F_NO_TIMING, F_BLOCKING_WAIT, F_NO_DEFER, F_ALWAYS_DEFER, F_SWEEP_TURNS, F_TIMING_SAMPLED, F_STREAM_SCREEN, F_SCREEN_ITEMS_1, F_SCREEN_ITEMS_2, F_NO_FUSED_SCREEN, F_SCREEN_WIDE, F_ONE_LAUNCH, run_device_batches, DeviceBatch,
    EXPORTED_SYMBOLS, CLASS_NAMES, CLASS_KERNELS, PinnedArray, Stream, StreamGroup, stream_device_of, HANDLE_ELSEWHERE, RecSink, OvlRec, OVL_REC_DTYPE,
)
