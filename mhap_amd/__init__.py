"""mhap_amd — MI355X-native MinHash overlap engine (the hot path of marbl/MHAP).

The compute lives in libmhaphip.so (hand-written HIP for gfx950 behind the C ABI in
include/mhap_hip.h).  This package is the Python host-side mirror of the reference's operator
interface (MinHashSearch / SequenceSketchStreamer / FastaData / MatchResult) over that C ABI.
There is no CPU fallback: without the built extension and a HIP device the compute calls raise.
"""
from .api import (  # noqa: F401
    MhapError,
    MhapParams,
    MinHashSearch,
    MinHashSearchGroup,
    FastaData,
    FastaScan,
    FrequencyCounts,
    MatchResult,
    format_record,
    synth_reads,
    records_to_lines,
    load_library,
    KERNEL_NAMES,
)
