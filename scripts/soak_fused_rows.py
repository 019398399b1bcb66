"""Randomised soak of the fused exact row top-k for big batches (api.hip: search_rows_fused_hi; candidate pass = maxsim_pp.hip MODE 2 with the
query fragments in registers and the records staged in LDS) against the dense path and the oracle.

    python scripts/soak_fused_rows.py [seconds] [seed]

`ORDER BY dist LIMIT k` for B >= 96 queries (`/root/reference/src/raglite/_search.py:69-79`): float data == the eight-group tile
(option fused_pp = 0) and == one round (fused_two_rounds = 0) bit for bit, integer data == the oracle (ties to the lowest row); random
corpus sizes / dims (256, 512, 1024), B 96..2100, k 1..512, cosine / dot, clustered corpora (many near-ties: long record logs)."""
from __future__ import annotations

import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))

import torch  # noqa: E402

import raglite_amd  # noqa: E402
from oracle import oracle  # noqa: E402


def main() -> None:
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    raglite_amd.set_device(0)
    t0 = time.time()
    cases = fallbacks = 0
    while time.time() - t0 < budget:
        metric = "cosine" if rng.random() < 0.5 else "dot"
        dim = int(rng.choice([256, 512, 1024]))
        n = int((64 << 20) // dim + rng.integers(1, 60_000))
        B = int(rng.choice([96, 97, 130, 512, 513, 1000, 2100]))
        k = int(rng.choice([1, 10, 100, 333, 512]))
        kind = "small_int" if rng.random() < 0.4 else "uniform"
        E = torch.empty((n, dim), dtype=torch.float32, device="cuda")
        raglite_amd.synth_fill(E, seed=int(rng.integers(1, 1 << 30)), kind=kind)
        Q = torch.empty((B, dim), dtype=torch.float32, device="cuda")
        raglite_amd.synth_fill(Q, seed=int(rng.integers(1, 1 << 30)), kind=kind)
        if rng.random() < 0.3:  # clustered: a few thousand rows near the queries' mean (long candidate lists, many records per wave)
            hot = torch.as_tensor(rng.choice(n, 3000, replace=False), device="cuda")
            E[hot] = Q.mean(dim=0, keepdim=True) + 1e-3 * torch.randn((3000, dim), device="cuda")
        idx = raglite_amd.DeviceIndex(E, metric=metric)
        S, R = idx.search_rows(Q, k)
        st = idx.filter_stats()
        # the yardstick: the eight-group tile when the candidate pass answered; the dense path when it gave up (a full list or record log --
        # e.g. cosines over 16-row blocks whose norms differ wildly, where the block-wide test of the sixteen-group tile passes too much):
        # the dense path's split-arithmetic sums differ from the exact re-scoring in the last bits, so only like compares with like
        with idx.options(**({"fused_topk": 0} if st["fallback"] else {"fused_pp": 0})):
            S0, R0 = idx.search_rows(Q, k)
            st0 = idx.filter_stats()
        if not st["fallback"] and st0["fallback"]:  # (the eight-group tile gave up where the sixteen-group one did not: dense again)
            with idx.options(fused_topk=0):
                S0, R0 = idx.search_rows(Q, k)
        assert torch.equal(R, R0) and torch.equal(S.view(torch.int32), S0.view(torch.int32)), ("pp != yardstick", metric, dim, n, B, k, kind, st, st0)
        with idx.options(fused_two_rounds=0):
            S1, R1 = idx.search_rows(Q, k)
            st1 = idx.filter_stats()
        if st1["fallback"] == st["fallback"]:
            assert torch.equal(R, R1) and torch.equal(S.view(torch.int32), S1.view(torch.int32)), ("two rounds != one", metric, dim, n, B, k, kind, st)
        if kind == "small_int" and metric == "dot":
            Eh = E.cpu().numpy()
            for b in (0, B - 1):
                _, rr = oracle.search_rows(Eh, Q[b].cpu().numpy(), k, metric, np.float64)
                assert np.array_equal(R[b].cpu().numpy(), rr), ("oracle", metric, dim, n, B, k)
        fallbacks += int(bool(st["fallback"]))
        cases += 1
        idx.close()
        del E, Q
    print(f"soak_fused_rows: {cases} cases ({fallbacks} through the dense fallback) in {time.time() - t0:.0f} s, all equal")


if __name__ == "__main__":
    main()
