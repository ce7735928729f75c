"""DESIGN.md is a design document (the current state, readable), not a log: <= 46 000 bytes and no line of DESIGN.md or
README.md beyond 200 characters (the history lives in docs/history/ and in git)."""
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_design_fits_and_lines_are_readable():
    assert os.path.getsize(os.path.join(ROOT, "DESIGN.md")) <= 46000
    for name in ("DESIGN.md", "README.md"):
        with open(os.path.join(ROOT, name), encoding="utf-8") as f:
            long_lines = [(i + 1, len(l.rstrip("\n"))) for i, l in enumerate(f) if len(l.rstrip("\n")) > 200]
        assert not long_lines, f"{name}: lines over 200 characters: {long_lines[:10]}"


def test_design_names_every_section_the_brief_asks_for():
    txt = open(os.path.join(ROOT, "DESIGN.md"), encoding="utf-8").read()
    for h in ("## 1. The path and its boundary", "## 3. Data layout in HBM", "## 4. Kernels", "## 5. Oracle and parity", "## 6. Measurement",
              "## 7. Multi-GPU", "## 8. Out of scope"):
        assert h in txt, h
