#!/usr/bin/env python3
"""Where does a block of the dW GEMM (gemm_tn_kernel<4,4>) spend its cycles?  Runs the diagnostic build
(python ideal-nerf_amd/build.py --variant diagtn "-DIDN_DIAG" train.hip) on the train workload and prints
per-category shares of block cycles (wave 0 of every block).  Shares only: the stamps cost cycles."""
import ctypes as C
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("IDN_LIB", os.path.join(ROOT, "ideal-nerf_amd", "libidealnerf_diagtn.so"))
import torch  # noqa: E402
import idealnerf_amd  # noqa: E402

lib = idealnerf_amd._lib.load()
lib.idealnerf_diag_tn_read.argtypes = [C.POINTER(C.c_ulonglong)]
sys.argv = ["bench.py", "--workload", "train", "--steps", "3", "--warmup", "1"]
import importlib.util  # noqa: E402
spec = importlib.util.spec_from_file_location("idn_bench", os.path.join(ROOT, "bench.py"))
bench = importlib.util.module_from_spec(spec)
spec.loader.exec_module(bench)
buf = (C.c_ulonglong * 8)()
lib.idealnerf_diag_tn_read(buf)   # zero
bench.main()
torch.cuda.synchronize()
lib.idealnerf_diag_tn_read(buf)
tot, wait, loop, epi, blocks, chunks = [buf[i] for i in range(6)]
print(json.dumps({"blocks": blocks, "chunks": chunks, "cycles_per_block": tot / max(blocks, 1),
                  "cycles_per_chunk_in_loop": loop / max(chunks, 1), "cycles_per_chunk_waiting": wait / max(chunks, 1),
                  "share_wait": wait / tot, "share_loop": loop / tot, "share_epilogue": epi / tot,
                  "share_prologue": (tot - wait - loop - epi) / tot,
                  "ideal_cycles_per_chunk_4x4": 8 * 16 * 64, "ideal_cycles_per_chunk_4x1": 8 * 4 * 64}))
