#!/usr/bin/env python3
"""Where does a 16-point chunk of the bf16-piece dW GEMM (gemm_tn_x6_kernel) spend its cycles?  Runs the diagnostic build
(python ideal-nerf_amd/build.py --variant diagx6 "-DIDN_DIAG_X6" train.hip) on the train workload and prints the mean cycles
of a chunk's four phases (wave 0 of every workgroup) next to their MFMA time: row 0 (24 MFMAs behind the chunk's LDS reads), rows
1..3 up to the chunk barrier (60 MFMAs, each followed by a slice of the next chunk's split / store / reload), the barrier, the
tail (12 MFMAs).  The stamps wait for the LDS queue (s_memtime returns through lgkmcnt): shares, not absolute times."""
import ctypes as C
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("IDN_LIB", os.path.join(ROOT, "ideal-nerf_amd", "libidealnerf_diagx6.so"))
import torch  # noqa: E402
import idealnerf_amd  # noqa: E402

lib = idealnerf_amd._lib.load()
lib.idealnerf_diag_x6_read.argtypes = [C.POINTER(C.c_ulonglong)]
sys.argv = ["bench.py", "--workload", "train", "--steps", "3", "--warmup", "1"]
import importlib.util  # noqa: E402
spec = importlib.util.spec_from_file_location("idn_bench", os.path.join(ROOT, "bench.py"))
bench = importlib.util.module_from_spec(spec)
spec.loader.exec_module(bench)
buf = (C.c_ulonglong * 8)()
lib.idealnerf_diag_x6_read(buf)   # zero
bench.main()
torch.cuda.synchronize()
lib.idealnerf_diag_x6_read(buf)
row0, rows, bar, tail, outside, chunks, wgs, total = [buf[i] for i in range(8)]
c = max(chunks, 1)
print(json.dumps({"workgroups": wgs, "chunks": chunks, "cycles_per_workgroup": total / max(wgs, 1),
                  "per_chunk": {"row0": row0 / c, "rows_1_3_to_barrier": rows / c, "barrier": bar / c, "tail": tail / c,
                                "sum": (row0 + rows + bar + tail) / c},
                  "mfma_cycles_per_chunk": {"row0": 24 * 32, "rows_1_3_to_barrier": 60 * 32, "tail": 12 * 32, "sum": 96 * 32},
                  "outside_the_loop_per_workgroup": outside / max(wgs, 1),
                  "share_outside_the_loop": outside / max(total, 1)}))
