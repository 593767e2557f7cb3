"""The documents name files of this repository in backquotes (profiles, logs, tests, sources):
every such path must exist, so that a reader who follows a reference finds something."""
import os
import re

import pytest

import psim_testlib as T

DOCS = ["DESIGN.md", "README.md", "INTEGRATION.md", "profiles/README.md", "scripts/README.md",
        "tests/golden/README.md", "rust/psim-sys/README.md"]
PATH = re.compile(r"`((?:profiles|tests|scripts|oracle|include|rust|particle-simulation-3d_amd|csrc)"
                  r"/[A-Za-z0-9_./\-]+)`")
# names that stand for a family of files, or for something that is deliberately absent
PLACEHOLDERS = {"oracle/_ref", "profiles/r03_bench_configN.json",
                "particle-simulation-3d_amd/csrc/libpsim_hip.so"}      # the last one is built, not kept


@pytest.mark.parametrize("doc", DOCS)
def test_paths_named_in_the_documents_exist(doc):
    path = os.path.join(T.ROOT, doc)
    if not os.path.exists(path):
        pytest.skip(f"{doc} is not part of this tree")
    missing = []
    with open(path, encoding="utf-8") as f:
        text = f.read()
    for m in PATH.finditer(text):
        name = m.group(1).rstrip(".")
        if name in PLACEHOLDERS:
            continue
        candidates = [name, os.path.join("particle-simulation-3d_amd", name)]
        if not any(os.path.exists(os.path.join(T.ROOT, c)) for c in candidates):
            missing.append(name)
    assert not missing, f"{doc} names files that do not exist: {sorted(set(missing))}"
