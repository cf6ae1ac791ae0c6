"""Run the diagnostic build of K12 with the tile body cut after phase <stop> (for rocprofv3 --pmc
instruction-count passes).  usage: pmc_phase.py <stop 0..4|99> [mode]"""
import ctypes as C, os, sys
import torch
sys.path.insert(0, ".")
so = "build/diag/libhbs_diag.so"
assert os.path.exists(so), "run `make diag` first"
import hevcbitstream_amd.api as api
api.library_path = lambda: so
import hevcbitstream_amd as hbs
stop = int(sys.argv[1]); mode = int(sys.argv[2]) if len(sys.argv) > 2 else 0
ctx = hbs.Context(0)
ctx.set_kernel(2)
lib = api.load_library()
g = ctx.synth_stream(0x1234, 104858, mode)
sb = g["stream_bytes"]
index, rbsp, summary, cap = ctx.alloc_outputs(sb, index_cap=104858 + 8)
assert lib.hbs_debug_set_stop(C.c_int(stop)) == 0
for _ in range(3):
    ctx.index_extract_async(g["stream"][:sb], index, cap, rbsp, summary)
torch.cuda.synchronize()
