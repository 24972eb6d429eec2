"""GPU-box helper: k_column_masks on one big view (config D root, 10k x 20k = 200 MB) for a few row-chunk sizes."""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import make_prg_amd.engine as E
from make_prg_amd.backend import HipBackend
from make_prg_amd.engine import BatchEngine, NodeRec
from make_prg_amd.msa import MSA

S, C = int(sys.argv[1]) if len(sys.argv) > 1 else 10000, int(sys.argv[2]) if len(sys.argv) > 2 else 20000
rng = np.random.default_rng(0)
data = np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, (S, C))]
msa = MSA(_data=data, _ids=[f"s{i}" for i in range(S)], _descs=[""] * S)
be = HipBackend(0)
eng = BatchEngine(be, 5, 7)
eng.load([msa])
nodes = [NodeRec(0, -1, 0, None, 0, C)]
tab, rowidx, total_cols, _ = eng._view_table(nodes, [0])
d_views, d_rowidx = be.upload(tab), be.upload(rowidx)
for rpc in (1024, 512, 256, 128, 64, 32):
    work, _ = BatchEngine._mask_work(tab, rpc)
    d_work, d_mask = be.upload(work), be.zeros(4 * total_cols)
    for rep in range(3):
        be.call("mprg_column_masks", be.ptr(eng.d_arena), be.ptr(d_views), be.ptr(d_rowidx), be.ptr(d_work), work.shape[0],
                rpc, be.ptr(d_mask), be.stream)
    be.synchronize()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 20
    ev0.record()
    for rep in range(n):
        be.call("mprg_column_masks", be.ptr(eng.d_arena), be.ptr(d_views), be.ptr(d_rowidx), be.ptr(d_work), work.shape[0],
                rpc, be.ptr(d_mask), be.stream)
    ev1.record()
    be.synchronize()
    ms = ev0.elapsed_time(ev1) / n
    print(f"rows_per_chunk {rpc:5d}  items {work.shape[0]:6d}  {ms:8.4f} ms  {S * C / ms * 1e-6:8.1f} GB/s", flush=True)
