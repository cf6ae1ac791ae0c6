"""hevcbitstream_amd -- MI355X (gfx950) Annex-B indexer / RBSP extractor.

Python is plumbing here: it loads the C-ABI library (include/hevcbitstream_amd.h,
built from csrc/*.hip by `make lib`) with ctypes and hands it device pointers of
torch tensors.  All work happens in the HIP kernels; there is no CPU fallback --
loading fails loudly when the library has not been built, and creating a
context fails when there is no gfx950 GPU."""
from .api import (Context, HbsError, NAL_ENTRY, PARSED, SUMMARY, ST_ERROR, ST_TRAILING03,  # noqa: F401
                  ST_UNTERMINATED, library_path, load_library, source_digest)

__all__ = ["Context", "HbsError", "NAL_ENTRY", "PARSED", "SUMMARY", "ST_ERROR", "ST_TRAILING03",
           "ST_UNTERMINATED", "library_path", "load_library", "source_digest"]
