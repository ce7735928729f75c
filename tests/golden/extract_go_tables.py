#!/usr/bin/env python3
"""Extract the reference's own golden vectors for the roaring hot path into JSON fixtures.

Runs ONLY in the build container (needs /root/reference, which does not exist on the GPU
box); the JSON it writes is committed next to it and is what the tests read.

    python tests/golden/extract_go_tables.py [/root/reference]

It does not copy code: it parses the *data literals* of Go table-driven tests
(roaring/roaring_internal_test.go) with a tiny Go-composite-literal parser and records
them with their source line, so that the oracle (oracle/roaring_oracle.c) can be pinned
to exactly the inputs/outputs the reference's tests pin.  How each table is applied
(which function, which fields) is restated by hand in tests/test_oracle_golden.py with
the reference line of the loop that consumes it.
"""
from __future__ import annotations

import json
import os
import re
import sys

REF = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))

# ------------------------------------------------------------------------------------
# Go literal tokenizer / parser (just enough for the test tables)
# ------------------------------------------------------------------------------------
TOKEN = re.compile(
    r"""
    (?P<ws>\s+|//[^\n]*|/\*.*?\*/)
  | (?P<num>0[xX][0-9a-fA-F_]+|\d[\d_]*)
  | (?P<str>"(?:[^"\\]|\\.)*")
  | (?P<id>[A-Za-z_][A-Za-z0-9_.]*)
  | (?P<op><<|>>|&\^|[{}()\[\],:+\-*/|&^~])
    """,
    re.X | re.S,
)

CONSTS = {
    "true": True,
    "false": False,
    "nil": None,
    "bitmapN": 1024,
    "MaxContainerVal": 0xFFFF,
    "ArrayMaxSize": 4096,
    "runMaxSize": 2048,
    "maxBitmap": 0xFFFFFFFFFFFFFFFF,
    "containerWidth": 65536,
}
TYPE_WORDS = {"uint16", "uint64", "int32", "int", "uint32", "int64", "uint", "byte", "Interval16", "bool", "string", "uint8"}


class ParseError(Exception):
    pass


def tokenize(src: str):
    pos, out = 0, []
    while pos < len(src):
        m = TOKEN.match(src, pos)
        if not m:
            raise ParseError(f"cannot tokenize at {src[pos:pos+30]!r}")
        pos = m.end()
        if m.lastgroup == "ws":
            continue
        out.append((m.lastgroup, m.group()))
    return out


class P:
    def __init__(self, toks):
        self.t, self.i = toks, 0

    def peek(self, k=0):
        return self.t[self.i + k] if self.i + k < len(self.t) else ("eof", "")

    def eat(self, val=None):
        kind, v = self.peek()
        if val is not None and v != val:
            raise ParseError(f"expected {val!r} got {v!r} at token {self.i}")
        self.i += 1
        return kind, v

    # value := [typeprefix] '{' elems '}' | expr
    def value(self):
        # type prefixes:  []T  [N]T  [bitmapN]T  T   followed by '{' or '('
        save = self.i
        if self.peek()[1] == "[":
            self.eat("[")
            while self.peek()[1] != "]":
                self.eat()
            self.eat("]")
            self.eat()  # element type
            if self.peek()[1] == "{":
                return self.composite()
            if self.peek()[1] == "(":  # []uint16(nil)
                self.eat("(")
                v = self.value()
                self.eat(")")
                return v
            self.i = save
        if self.peek()[0] == "id" and self.peek()[1] in TYPE_WORDS and self.peek(1)[1] == "{":
            self.eat()
            return self.composite()
        if self.peek()[1] == "{":
            return self.composite()
        # helper call: MakeBitmap([]uint64{..}), bitmapOddBitsSet(), make([]uint64, bitmapN)...
        # recorded symbolically; tests/golden/go_fixtures.py restates what each helper builds
        if (
            self.peek()[0] == "id"
            and self.peek(1)[1] == "("
            and self.peek()[1] not in TYPE_WORDS
            and self.peek()[1] not in CONSTS
        ):
            name = self.eat()[1]
            self.eat("(")
            args = []
            if name == "make":  # make([]T, n): keep only the length
                while self.peek()[1] != ",":
                    self.eat()
                self.eat(",")
            while self.peek()[1] != ")":
                args.append(self.value())
                if self.peek()[1] == ",":
                    self.eat(",")
            self.eat(")")
            return {"$call": name, "args": args}
        return self.expr()

    def composite(self):
        self.eat("{")
        items, is_dict, d = [], False, {}
        while self.peek()[1] != "}":
            if self.peek()[0] == "id" and self.peek(1)[1] == ":" and self.peek()[1] not in CONSTS:
                key = self.eat()[1]
                self.eat(":")
                d[key] = self.value()
                is_dict = True
            else:
                items.append(self.value())
            if self.peek()[1] == ",":
                self.eat(",")
        self.eat("}")
        return d if is_dict else items

    # expr := unary { binop unary }   (left-assoc, Go precedence approximated by python eval)
    def expr(self):
        parts = []
        depth = 0
        while True:
            kind, v = self.peek()
            if kind == "eof":
                break
            if depth == 0 and v in {",", "}", ":", "]"}:
                break
            if depth == 0 and v == ")":
                break
            if v == "(":
                depth += 1
            elif v == ")":
                depth -= 1
            if v == "{":
                raise ParseError("composite inside expression")
            self.eat()
            if kind == "num":
                parts.append(str(int(v.replace("_", ""), 0)))
            elif kind == "id":
                if v in CONSTS:
                    parts.append(repr(CONSTS[v]))
                elif v in TYPE_WORDS:  # conversion: uint16(x) -> (x)
                    parts.append("")
                else:
                    raise ParseError(f"unknown identifier {v}")
            elif kind == "str":
                parts.append(v)
            elif v == "&^":
                parts.append("&~")
            elif v == "^":
                # unary ^x is bitwise not in Go; binary ^ is xor
                prev = parts[-1] if parts else ""
                parts.append("^" if prev and (prev[-1].isalnum() or prev[-1] == ")") else "~")
            else:
                parts.append(v)
        text = " ".join(parts).strip()
        if text == "":
            raise ParseError("empty expression")
        val = eval(text, {"__builtins__": {}})  # numbers / strings / None / bools only
        if isinstance(val, int) and not isinstance(val, bool) and val < 0 and "~" in text:
            val &= 0xFFFFFFFFFFFFFFFF
        return val


def parse_value(src: str):
    p = P(tokenize(src))
    v = p.value()
    if p.peek()[0] != "eof":
        raise ParseError(f"trailing tokens: {p.t[p.i:p.i+5]}")
    return v


# ------------------------------------------------------------------------------------
# locating tables
# ------------------------------------------------------------------------------------
def matching_brace(src: str, open_idx: int) -> int:
    depth, i, n = 0, open_idx, len(src)
    while i < n:
        c = src[i]
        if c == '"':
            i += 1
            while src[i] != '"':
                i += 2 if src[i] == "\\" else 1
        elif c == "/" and src[i + 1] == "/":
            i = src.index("\n", i)
            continue
        elif c == "{":
            depth += 1
        elif c == "}":
            depth -= 1
            if depth == 0:
                return i
        i += 1
    raise ParseError("unbalanced braces")


def func_body(src: str, name: str):
    m = re.search(r"^func %s\(" % re.escape(name), src, re.M)
    if not m:
        raise ParseError(f"func {name} not found")
    ob = src.index("{", m.end())
    cb = matching_brace(src, ob)
    return ob, cb


def line_of(src: str, idx: int) -> int:
    return src.count("\n", 0, idx) + 1


def struct_table(src: str, test: str, var: str = "tests"):
    """First `<var> := []struct { fields } { rows }` inside func `test`."""
    ob, cb = func_body(src, test)
    m = re.compile(r"\b%s\s*:=\s*\[\]struct\s*\{" % re.escape(var)).search(src, ob, cb)
    if not m:
        raise ParseError(f"{test}: no table {var}")
    s_open = m.end() - 1
    s_close = matching_brace(src, s_open)
    fields = []
    for ln in src[s_open + 1 : s_close].splitlines():
        ln = ln.split("//")[0].strip()
        if not ln:
            continue
        names = ln.split()[0]
        # "a, b []uint16" style
        head = ln[: ln.rfind(" ")] if " " in ln else ln
        for nm in head.replace(",", " ").split():
            if re.match(r"^[A-Za-z_]\w*$", nm) and nm not in TYPE_WORDS:
                fields.append(nm)
        del names
    r_open = src.index("{", s_close + 1)
    r_close = matching_brace(src, r_open)
    rows = parse_value(src[r_open : r_close + 1])
    out = []
    for r in rows:
        if isinstance(r, list):
            r = dict(zip(fields, r))
        out.append(r)
    return {"line": line_of(src, m.start()), "fields": fields, "rows": out}


def main():
    path = os.path.join(REF, "roaring", "roaring_internal_test.go")
    src = open(path).read()

    # ---- 1. TestContainerCombinations (roaring_internal_test.go:2974-3771)
    ob, cb = func_body(src, "TestContainerCombinations")
    body = src[ob:cb]
    combos = []
    for m in re.finditer(r'^\s*\{(\w+),\s*"(\w*)",\s*"(\w*)",\s*"(\w*)"\},?\s*$', body, re.M):
        combos.append({"op": m.group(1), "x": m.group(2), "y": m.group(3), "exp": m.group(4), "line": line_of(src, ob + m.start(1))})
    with open(os.path.join(OUT, "container_combinations.json"), "w") as f:
        json.dump(
            {
                "source": "roaring/roaring_internal_test.go:2974-3771 TestContainerCombinations; fixtures roaring/roaring_helpers_test.go:12-306",
                "ops": combos,
            },
            f,
            indent=0,
        )
    print(f"container_combinations.json: {len(combos)} triples")

    # ---- 2. per-kernel table tests
    tables = [
        "TestRunAppendInterval",
        "TestBitmapCountRange",
        "TestIntersectionCountArrayBitmap2",
        "TestIntersectionCountRunRun",
        "TestIntersectArrayRun",
        "TestIntersectRunRun",
        "TestIntersectBitmapRunBitmap",
        "TestIntersectBitmapRunArray",
        "TestUnionInterval16InPlace",
        "TestUnionRunRun",
        "TestUnionArrayRun",
        "TestBitmapSetRange",
        "TestArrayToBitmap",
        "TestBitmapToArray",
        "TestRunToBitmap",
        "TestBitmapToRun",
        "TestArrayToRun",
        "TestRunToArray",
        "TestBitmapZeroRange",
        "TestUnionBitmapRun",
        "TestBitmapCountRuns",
        "TestArrayCountRuns",
        "TestDifferenceArrayRun",
        "TestDifferenceRunArray",
        "TestDifferenceRunBitmap",
        "TestDifferenceBitmapRun",
        "TestDifferenceBitmapArray",
        "TestDifferenceBitmapBitmap",
        "TestDifferenceRunRun",
        "TestXorArrayRun",
        "TestXorRunRun",
        "TestBitmapXorRange",
        "TestXorBitmapRun",
        "TestIntersectArrayBitmap",
    ]
    out, skipped = {}, {}
    for t in tables:
        try:
            out[t] = struct_table(src, t)
        except (ParseError, SyntaxError, ValueError, NameError, TypeError) as e:  # noqa: PERF203
            skipped[t] = str(e)
    with open(os.path.join(OUT, "roaring_internal_tables.json"), "w") as f:
        json.dump({"source": "roaring/roaring_internal_test.go (table-driven tests; `line` = first line of each table)", "tables": out}, f)
    for t, tb in out.items():
        print(f"  {t}: {len(tb['rows'])} rows, fields {tb['fields']}")
    for t, e in skipped.items():
        print(f"  SKIPPED {t}: {e}")


if __name__ == "__main__":
    main()
