#!/usr/bin/env python3
"""Where does the device's inverse-CDF put the boundary between two neighbouring candidates?  (GPU box; debug tool.)

The sampler kernel draws `u` against a cumulative distribution of 2^-40 fixed-point masses of __expf; the oracle
(oracle/ref_sample.py) uses float64 probabilities.  This probe locates, by a 64-ary search on `u` (one launch = 64 copies
of the same logits row with 64 different uniforms), the device's edge between candidates j and j+1 for a few j per row
and prints |device edge - oracle edge| next to what correctly rounded float32 arithmetic alone would give.  It is the
measurement behind the bound asserted in tests/test_gpu_kernels.py::test_sampler_top_p_injected_uniforms.

    python tools/debug/sampler_edge_probe.py [--rows 4]
"""
from __future__ import annotations

import argparse
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))

import torch  # noqa: E402

from mlx_parallm_amd import _lib as L  # noqa: E402
from oracle import ref_sample  # noqa: E402  (debug tool: the oracle is the checker here)
from oracle.numerics import round_to  # noqa: E402

NB = 64


def run(lg_row, temp, top_p, us):
    B, V = len(us), lg_row.shape[0]
    t = torch.from_numpy(np.repeat(lg_row[None], B, 0).copy()).cuda()
    ud = torch.from_numpy(np.asarray(us, np.float32)).cuda()
    toks = torch.zeros(B, dtype=torch.int32, device="cuda")
    lp = torch.zeros(B, dtype=torch.float32, device="cuda")
    p0 = torch.zeros(B, dtype=torch.float32, device="cuda")
    ki = torch.zeros((B, 1), dtype=torch.int32, device="cuda")
    kl = torch.zeros((B, 1), dtype=torch.float32, device="cuda")
    st = torch.zeros((B, 2), dtype=torch.float32, device="cuda")
    torch.cuda.synchronize()
    L.check(L.lib().mi_op_sample(t.data_ptr(), B, V, temp, top_p, ud.data_ptr(), 0, toks.data_ptr(), lp.data_ptr(),
                                 p0.data_ptr(), ki.data_ptr(), kl.data_ptr(), st.data_ptr()))
    return toks.cpu().numpy()


def f32_model_edges(lg, temp, order):
    """The cumulative edges correctly rounded float32 arithmetic would give (floor(exp2(f32 arg) * 2^40))."""
    mx = lg.max()
    a = ((lg - mx).astype(np.float32) * (np.float32(1.0) / np.float32(temp))).astype(np.float32)
    e = np.exp2((a * np.float32(1.4426950408889634)).astype(np.float32).astype(np.float64)).astype(np.float32)
    m = np.floor(e.astype(np.float64) * 2.0 ** 40)[order]
    return np.cumsum(m) / m.sum()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=3)
    a = ap.parse_args()
    rng = np.random.default_rng(11)
    worst = 0.0
    for V, std, bf in ((32000, 1.3, True), (151936, 3.0, False), (5000, 3.0, False)):
        for temp, top_p in ((1.0, 0.9), (0.7, 0.5), (1.3, 1.0), (1.0, 0.05)):
            for r in range(a.rows):
                lg = (rng.standard_normal(V) * std).astype(np.float32)
                if bf:
                    lg = round_to(lg, "bfloat16")
                if top_p < 1.0:
                    ids, pr = ref_sample.top_p_candidates(lg, top_p, temp)
                else:
                    x = lg.astype(np.float64) / temp
                    p = np.exp(x - x.max())
                    ids = np.lexsort((np.arange(V), -p))
                    pr = p[ids]
                ids = np.asarray(ids)
                pr = np.asarray(pr, np.float64)
                cum = np.cumsum(pr / pr.sum())
                # float32 model restricted to the same candidates
                cm = f32_model_edges(lg, temp, ids)
                n = len(ids)
                # probe a few boundaries spread over the candidate list; uniforms are float32 on the device, so the search
                # resolution is 2^-24 .. 2^-25
                for j in sorted(set(int(q * (n - 1)) for q in (0.0, 0.1, 0.3, 0.5, 0.7, 0.9, 0.97))):
                    if j >= n - 1 or ids[j] == ids[j + 1]:
                        continue
                    lo, hi = max(cum[j] - 4e-5, 0.0), min(cum[j] + 4e-5, 1.0 - 1e-7)
                    ok = True
                    for _ in range(4):
                        us = np.linspace(lo, hi, NB)
                        toks = run(lg, temp, top_p, us)
                        pos = {int(t): i for i, t in enumerate(ids.tolist())}
                        rank = np.array([pos.get(int(t), -1) for t in toks])
                        le = np.where(rank <= j)[0]
                        gt = np.where(rank > j)[0]
                        if len(le) == 0 or len(gt) == 0 or le.max() > gt.min():
                            ok = False
                            break
                        lo, hi = us[le.max()], us[gt.min()]
                    if not ok:
                        print(f"V {V} T {temp} top_p {top_p} row {r} boundary {j}: search window missed (ranks {rank[:4]}..{rank[-4:]})")
                        continue
                    edge = 0.5 * (lo + hi)
                    err = abs(edge - cum[j])
                    worst = max(worst, err)
                    print(f"V {V:6d} T {temp} top_p {top_p} row {r} boundary {j:6d}/{n:6d} at c = {cum[j]:.6f}: device edge - oracle "
                          f"{edge - cum[j]:+.3e} (search width {hi - lo:.1e}); float32 model - oracle {cm[j] - cum[j]:+.3e}", flush=True)
    print(f"worst |device edge - oracle edge| = {worst:.3e}")


if __name__ == "__main__":
    main()
