"""Generate `tests/golden/query_adapter.npz` by executing the REFERENCE's own query-adapter arithmetic
-- TEST INFRASTRUCTURE.  Run in the authoring container only (needs /root/reference):

    python -m oracle.make_golden_adapter

`src/raglite/_query_adapter.py` cannot be imported here (sqlmodel / tqdm / the database layer are absent), and its
arithmetic sits inside one function that also talks to the store.  So the REAL source text is executed piecewise,
nothing is copied into this repository:

* `_optimize_query_target` (`_query_adapter.py:20-38`): the function definition is cut out with `ast` and exec'd with
  the real `numpy` and `scipy.optimize.lsq_linear`;
* the closed-form block that turns the stacked (Q, T) into the adapter (`:182-205`: row normalisation, M = TᵀQ / n,
  null-space completion, Procrustes / Frobenius solution): the source lines between its first and last comment are
  exec'd on our Q, T and a stand-in `config`;
* positive / negative row selection (`:170-181`): `E[[np.argmax(E @ q)]]`, evaluated literally.
"""

from __future__ import annotations

import ast
import textwrap
import types
from pathlib import Path

import numpy as np
from scipy.optimize import lsq_linear

SRC = Path("/root/reference/src/raglite/_query_adapter.py")
OUT = Path(__file__).resolve().parent.parent / "tests" / "golden" / "query_adapter.npz"


def _reference_pieces():
    text = SRC.read_text()
    tree = ast.parse(text)
    fn = next(n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == "_optimize_query_target")
    fn_src = "\n".join(text.splitlines()[fn.lineno - 1 : fn.end_lineno])
    ns = {"np": np, "lsq_linear": lsq_linear, "FloatVector": np.ndarray, "FloatMatrix": np.ndarray}
    exec(compile(fn_src, str(SRC), "exec"), ns)  # noqa: S102 - the reference's own code, read-only
    lines = text.splitlines()
    a = next(i for i, ln in enumerate(lines) if "# Normalise the rows of Q and T." in ln)
    b = next(i for i, ln in enumerate(lines) if "# Store the optimal query adapter in the database." in ln)
    block = textwrap.dedent("\n".join(lines[a:b]))

    def adapter_from(Q, T, metric):  # noqa: N803
        env = {"np": np, "Q": Q.copy(), "T": T.copy(), "FloatMatrix": np.ndarray,
               "config": types.SimpleNamespace(vector_search_distance_metric=metric)}
        exec(compile(block, str(SRC), "exec"), env)  # noqa: S102
        return env["A_star"], env["Q"], env["T"]

    return ns["_optimize_query_target"], adapter_from


def main() -> None:
    optimize, adapter_from = _reference_pieces()
    rng = np.random.default_rng(2024)
    out: dict[str, np.ndarray] = {}
    # ---- target optimisation cases (fp16 inputs like the reference's embeddings) --------------------------
    n_t = 0
    for d, n_pos, n_neg, alpha in [(16, 1, 3, 0.05), (32, 2, 5, 0.05), (32, 4, 1, 0.2), (64, 3, 7, 0.05), (24, 1, 1, 0.0)]:
        q = rng.standard_normal(d)
        q = (q / np.linalg.norm(q)).astype(np.float16)
        P = rng.standard_normal((n_pos, d))
        N = rng.standard_normal((n_neg, d))
        P = (P / np.linalg.norm(P, axis=1, keepdims=True)).astype(np.float16)
        N = (N / np.linalg.norm(N, axis=1, keepdims=True)).astype(np.float16)
        t = optimize(q, P, N, α=alpha)
        for k, v in (("q", q), ("P", P), ("N", N), ("alpha", np.float64(alpha)), ("t", t)):
            out[f"target{n_t}_{k}"] = v
        n_t += 1
    out["n_target_cases"] = np.int64(n_t)
    # ---- adapter from stacked (Q, T): under- and over-determined, both metrics -----------------------------
    n_a = 0
    for d, n, metric in [(16, 5, "cosine"), (16, 40, "cosine"), (24, 7, "dot"), (24, 60, "dot"), (32, 32, "cosine")]:
        Q = rng.standard_normal((n, d))
        T = Q + 0.1 * rng.standard_normal((n, d))
        A, Qn, Tn = adapter_from(Q, T, metric)
        out[f"adapter{n_a}_Q"], out[f"adapter{n_a}_T"], out[f"adapter{n_a}_A"] = Q, T, A
        out[f"adapter{n_a}_metric"] = np.asarray(metric)
        n_a += 1
    out["n_adapter_cases"] = np.int64(n_a)
    # ---- positive / negative row selection ----------------------------------------------------------------------
    E = rng.standard_normal((9, 16)).astype(np.float16)
    q = rng.standard_normal(16).astype(np.float16)
    out["select_E"], out["select_q"] = E, q
    out["select_row"] = E[[np.argmax(E @ q)]]
    # ---- reciprocal rank fusion (`src/raglite/_search.py:233-252`): the function text, exec'd ----------------------
    stext = (SRC.parent / "_search.py").read_text()
    stree = ast.parse(stext)
    rrf = next(n for n in stree.body if isinstance(n, ast.FunctionDef) and n.name == "reciprocal_rank_fusion")
    rrf_src = "\n".join(stext.splitlines()[rrf.lineno - 1 : rrf.end_lineno])
    from collections import defaultdict

    ns2 = {"defaultdict": defaultdict, "ChunkId": str}
    exec(compile(rrf_src, str(SRC.parent / "_search.py"), "exec"), ns2)  # noqa: S102
    cases = [
        ([["a", "b", "c", "d"], ["c", "a", "e"]], None, 60),
        ([["a", "b", "c", "d"], ["c", "a", "e"]], [0.75, 0.25], 60),
        ([["x1", "x2"], []], [0.75, 0.25], 60),
        ([[], []], None, 60),
        ([[f"id{i}" for i in range(20)], [f"id{(i * 7) % 23}" for i in range(20)], [f"id{i}" for i in range(19, -1, -1)]],
         [1.0, 0.5, 0.25], 10),
    ]
    import json as _json

    rrf_out = []
    for rankings, weights, kk in cases:
        ids, sc = ns2["reciprocal_rank_fusion"](rankings, k=kk, weights=weights)
        rrf_out.append({"rankings": rankings, "weights": weights, "k": kk, "ids": ids, "scores": sc})
    out["rrf_json"] = np.asarray(_json.dumps(rrf_out))
    # ---- a5 + the num_hits rule, straight from vector_search's body (`src/raglite/_search.py:57-67`) ----------------
    slines = stext.splitlines()
    a5_a = next(i for i, ln in enumerate(slines) if "# Apply the query adapter to the query embedding." in ln)
    a5_b = next(i for i, ln in enumerate(slines) if "# Rank the chunks by relevance according to the L" in ln)
    a5_src = textwrap.dedent("\n".join(slines[a5_a:a5_b]))
    nh_a = next(i for i, ln in enumerate(slines) if "corrected_oversample = oversample * config.chunk_max_size" in ln)
    nh_src = textwrap.dedent("\n".join(slines[nh_a : nh_a + 2]))
    n_a5 = 0
    for d, dtype in [(16, np.float16), (64, np.float16), (256, np.float16), (32, np.float32)]:
        A = np.linalg.qr(rng.standard_normal((d, d)))[0]  # an orthogonal adapter, fp64 as stored (`_query_adapter.py:203-205`)
        q = rng.standard_normal(d)
        q = (q / np.linalg.norm(q)).astype(dtype)

        class _IndexMetadata:
            @staticmethod
            def get(id_="default", *, config=None):  # noqa: ANN001,ANN205
                return {"query_adapter": A}

        env = {"np": np, "IndexMetadata": _IndexMetadata, "query_embedding": q.copy(),
               "config": types.SimpleNamespace(vector_search_query_adapter=True)}
        exec(compile(a5_src, "_search.py", "exec"), env)  # noqa: S102
        out[f"a5_{n_a5}_A"], out[f"a5_{n_a5}_q"], out[f"a5_{n_a5}_out"] = A, q, env["query_embedding"]
        n_a5 += 1
    out["n_a5_cases"] = np.int64(n_a5)
    nh = []
    for oversample, chunk_max_size, num_results in [(4, 2048, 3), (4, 2048, 40), (4, 1024, 8), (2, 2048, 10), (4, 4096, 5), (3, 1536, 7)]:
        env = {"oversample": oversample, "num_results": num_results,
               "config": types.SimpleNamespace(chunk_max_size=chunk_max_size),
               "RAGLiteConfig": types.SimpleNamespace(chunk_max_size=2048)}  # the class default, `_config.py`
        exec(compile(nh_src, "_search.py", "exec"), env)  # noqa: S102
        nh.append([oversample, chunk_max_size, num_results, int(env["num_hits"])])
    out["num_hits_cases"] = np.asarray(nh, dtype=np.int64)
    np.savez(OUT, **out)
    print(f"wrote {OUT} ({n_t} target cases, {n_a} adapter cases)")


if __name__ == "__main__":
    main()
