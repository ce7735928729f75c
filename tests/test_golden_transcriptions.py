"""The two sets of reference vectors that are restated by hand (their Go tests build part of the data with loops)
against the literals extracted mechanically from the same tests (tests/golden/extract_literal_vectors.py ->
literal_vectors.json): every literal the transcription contains must be the extracted one."""
import json
import os

import go_bitmap_vectors as V

LIT = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "literal_vectors.json")))


def test_intersection_count_literals_match_the_reference_source():
    ic = LIT["intersection_count"]
    cases = {name: (bm0, bm1, exp) for name, bm0, bm1, exp in V.CASES}
    for name in ("ArrayArray", "ArrayRun", "RunRun"):  # both operands are literals
        lits, exp, opt = ic[name]["new_file_bitmap_literals"], ic[name]["expected_counts"], ic[name]["optimized"]
        (v0, o0), (v1, o1), e = cases[name]
        assert [v0, v1] == lits and exp == [e, e], name
        assert (o0, o1) == ("bm0" in opt, "bm1" in opt), name
    # one literal operand, the other built by a loop (restated in go_bitmap_vectors.py)
    assert cases["BitmapRun"][1][0] == ic["BitmapRun"]["new_file_bitmap_literals"][1] and ic["BitmapRun"]["expected_counts"] == [4, 4]
    assert cases["BitmapRun"][1][1] and ic["BitmapRun"]["optimized"] == ["bm1"]
    assert cases["ArrayBitmap"][0][0] == ic["ArrayBitmap"]["new_file_bitmap_literals"][0] and ic["ArrayBitmap"]["expected_counts"] == [3, 3]
    assert ic["BitmapBitmap"]["expected_counts"] == [2, 2] and cases["BitmapBitmap"][2] == 2
    assert [cases["Mixed/1"][1][0], cases["Mixed/3"][1][0]] == ic["Mixed"]["new_file_bitmap_literals"]
    assert [cases["Mixed/1"][2], cases["Mixed/3"][2]] == ic["Mixed"]["expected_counts"]


def test_bsi_add_cases_are_the_reference_literals():
    cases = LIT["bsi_add_cases"]
    assert len(cases) == 2 and len(cases[0]["positions"]) == 20 and max(cases[0]["b"]) == 9023592401
    for c in cases:
        assert len(c["positions"]) == len(c["a"]) == len(c["b"])
