"""The library variant the timing experiments need: libfbk.so built with -DFBK_EXPERIMENTS (options pair_ablate, pair_stamp,
matrix_fused_ablate and the device branches behind them — WRONG results by design, which is why the product library has
none of it).  Built into build_variants/experiments/ (git-ignored; travels to the GPU box like the product .so).

    import _experiments; _experiments.use()     # BEFORE importing featurebase_amd

`use()` builds the variant when it is missing or older than its sources and exports FBK_LIB_PATH, which
featurebase_amd/lib.py reads at import."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "featurebase_amd", "csrc")
OUT = os.path.join(ROOT, "build_variants", "experiments", "libfbk.so")


def build() -> str:
    sys.path.insert(0, ROOT)
    import __graft_entry__ as G

    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hip", ".h", ".inc"))] + [os.path.join(ROOT, "include", "fbk.h")]
    if not os.path.exists(OUT) or any(os.path.getmtime(d) > os.path.getmtime(OUT) for d in deps):
        os.makedirs(os.path.dirname(OUT), exist_ok=True)
        cmd = [G._hipcc()] + G.HIP_FLAGS + ["-DFBK_EXPERIMENTS", os.path.join(CSRC, "fbk.hip"), "-o", OUT]
        print("[experiments]", " ".join(cmd), file=sys.stderr, flush=True)
        subprocess.check_call(cmd, cwd=CSRC)
    return OUT


def use() -> str:
    if "featurebase_amd.lib" in sys.modules:
        raise RuntimeError("_experiments.use() must run before featurebase_amd is imported")
    os.environ["FBK_LIB_PATH"] = build()
    return os.environ["FBK_LIB_PATH"]


if __name__ == "__main__":
    print(build())
