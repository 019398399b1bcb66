"""Golden strings for `str(chunk)` from the reference's OWN code (test infrastructure; authoring container only).

The properties `Chunk.front_matter` / `Chunk.content` (/root/reference/src/raglite/_database.py:300-320) are cut out of the
module's source text with `ast` and compiled into a bare stand-in class (the module itself needs sqlmodel, absent here), then
evaluated on metadata shaped as the reference stores it -- every value a list (`_adapt_metadata`, _database.py:51-55).
Output: tests/golden/chunk_text.json, compared with raglite_amd._store.chunk_text by tests/test_oracle_golden.py."""

from __future__ import annotations

import ast
import json
from pathlib import Path

REF = Path("/root/reference/src/raglite/_database.py")
OUT = Path(__file__).resolve().parent.parent / "tests" / "golden" / "chunk_text.json"

CASES = [
    {"headings": "# Title\n\n## Section", "body": "Body text.", "metadata": {"filename": ["a.md"], "url": [None], "topic": ["t0"]}},
    {"headings": "  # Padded  ", "body": "\nbody\n", "metadata": {"filename": ["b.pdf"], "url": ["https://x.y/z"], "uri": ["s3://k"]}},
    {"headings": "", "body": "no front matter", "metadata": {}},
    {"headings": "# H", "body": "empty lists are falsy", "metadata": {"filename": [], "url": []}},
    {"headings": "# H", "body": "scalar values as a caller might pass them", "metadata": {"filename": "c.md", "url": None}},
    {"headings": "# H", "body": "two values", "metadata": {"filename": ["d.md", "e.md"], "size": [3]}},
]


def reference_chunk_class():
    tree = ast.parse(REF.read_text())
    cls = next(n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == "Chunk")
    keep = [n for n in cls.body if isinstance(n, ast.FunctionDef) and n.name in ("front_matter", "content", "__str__")]
    assert [n.name for n in keep] == ["front_matter", "content", "__str__"], "reference layout changed"
    mod = ast.Module(body=[ast.ClassDef(name="Chunk", bases=[], keywords=[], body=keep, decorator_list=[])], type_ignores=[])
    ast.fix_missing_locations(mod)
    ns: dict = {}
    exec(compile(mod, str(REF), "exec"), ns)  # noqa: S102 - the reference's own statements
    return ns["Chunk"]


def main() -> None:
    Chunk = reference_chunk_class()
    out = []
    for case in CASES:
        c = Chunk()
        c.headings, c.body, c.metadata_ = case["headings"], case["body"], case["metadata"]
        out.append({**case, "text": str(c)})
    OUT.write_text(json.dumps(out, indent=1) + "\n")
    print(f"wrote {OUT} ({len(out)} cases)")


if __name__ == "__main__":
    main()
