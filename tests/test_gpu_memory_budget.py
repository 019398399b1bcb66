"""The optional corpus images of a DeviceIndex (pre-split image, image of the hi halves, HI plane: 4 + 2 + 2 bytes per element next to the
rows) are built only while they leave headroom on the device (raglite_amd/csrc/api.hip: image_fits, `rl_index_memory`); an index that gets
none of them answers the same calls through the kernels over the stored rows with the same results -- the a6-a9 paths of
`/root/reference/src/raglite/_search.py:54-153` do not depend on which accelerator was affordable."""

import json
import os
import subprocess
import sys
from pathlib import Path

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = Path(__file__).resolve().parent.parent

CHILD = r"""
import json, sys
import numpy as np, torch
import raglite_amd
raglite_amd.set_device(0)
n, dim, nq, B, k = 70_000, 1024, 32, 5, 20
E = torch.empty((n, dim), dtype=torch.float32, device="cuda"); raglite_amd.synth_fill(E, seed=3, kind="small_int")
Q = torch.empty((B, nq, dim), dtype=torch.float32, device="cuda"); raglite_amd.synth_fill(Q, seed=4, kind="small_int")
rng = np.random.default_rng(1)
sizes = rng.integers(1, 16, n); off = np.concatenate(([0], np.cumsum(sizes))); off = off[off <= n]
if off[-1] != n: off = np.concatenate((off, [n]))
idx = raglite_amd.DeviceIndex(E, off.astype(np.int64), metric="dot")
mem = idx.memory()
s, c = idx.maxsim_topk_batch(Q, k)
rs, rr = idx.search_rows(Q[:, 0, :].contiguous(), k)
print(json.dumps({"mem": mem, "scores": s.cpu().numpy().tolist(), "chunks": c.cpu().numpy().tolist(),
                  "row_scores": rs.cpu().numpy().tolist(), "rows": rr.cpu().numpy().tolist(), "filter": idx.filter_stats()["kind"]}))
"""


def _run(extra_env):
    env = dict(os.environ, **extra_env)
    res = subprocess.run([sys.executable, "-c", CHILD], cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stderr[-2000:]
    return json.loads(res.stdout.strip().splitlines()[-1])


def test_index_without_room_for_its_images_gives_the_same_results():
    with_images = _run({})
    m = with_images["mem"]
    assert m["rows"] == 70_000 * 1024 * 4
    assert m["presplit_image"] >= m["rows"] and m["hi_image"] >= m["rows"] // 2 and m["hi_plane"] >= m["rows"] // 2
    assert m["device_total"] > m["device_free"] > 0 and m["image_headroom"] >= 2 << 30
    assert with_images["filter"] in ("maxsim_batch_hi", "rows_hi")
    # a headroom nobody can leave: no image is built, every call runs over the stored rows
    without = _run({"RAGLITE_IMAGE_HEADROOM_MB": str(1 << 30)})
    w = without["mem"]
    assert w["presplit_image"] == 0 and w["hi_image"] == 0 and w["hi_plane"] == 0 and w["rows"] == m["rows"]
    assert without["filter"] == "none"
    # integer-valued data: every path is exact, so the bits agree
    assert without["chunks"] == with_images["chunks"] and without["scores"] == with_images["scores"]
    assert without["rows"] == with_images["rows"] and without["row_scores"] == with_images["row_scores"]
