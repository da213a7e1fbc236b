"""Builds libcnerf_hip.so (gfx950) in-tree with hipcc.  `python -m consistentnerf_amd.build [-f]`.

hipcc cross-compiles without a GPU; the resulting .so travels to the GPU box with the repo snapshot.
Objects are rebuilt only when a source/header is newer (or with -f)."""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
# CN_BUILD_TAG=<name> builds an experiment variant (with CN_EXTRA_FLAGS) next to the product library; a variant is
# only ever loaded when CNERF_LIB_PATH points at it (scripts/kvariants.sh).
TAG = os.environ.get("CN_BUILD_TAG", "")
OBJ = os.path.join(CSRC, "build" + ("_" + TAG if TAG else ""))
LIB = os.path.join(HERE, "libcnerf_hip.so") if not TAG else os.path.join(ROOT, "variants", f"libcnerf_{TAG}.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
# -ffp-contract=off: the reference composes separately-rounded ATen ops; FMA contraction would change
# sample positions / encodings by an ulp that 2^9-frequency encodings amplify.  MFMA code is unaffected.
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off",
         "-I" + os.path.join(ROOT, "include"), "-I" + CSRC, "-Wno-unused-result"] + os.environ.get("CN_EXTRA_FLAGS", "").split()


def sources():
    return sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))


def _stale(out, deps):
    if not os.path.exists(out):
        return True
    t = os.path.getmtime(out)
    return any(os.path.getmtime(d) > t for d in deps)


def _check_ticket_isa(verbose):
    """composite.hip publishes a per-workgroup partial and then takes a ticket; correctness rests on an `s_waitcnt vmcnt(0)` between
    the two in the generated ISA (a compiler-version property): asserted HERE, whenever the file is (re)compiled, so that a toolchain
    that reorders them cannot produce a library at all (the check used to live only in a skippable test)."""
    import importlib.util
    asm = os.path.join(OBJ, "composite.s")
    cmd = [HIPCC] + FLAGS + ["--cuda-device-only", "-S", os.path.join(CSRC, "composite.hip"), "-o", asm]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("hipcc -S failed on composite.hip:\n" + r.stdout + r.stderr)
    spec = importlib.util.spec_from_file_location("isa_ticket_check", os.path.join(ROOT, "scripts", "isa_ticket_check.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    bad = mod.check(asm, need_sites=6)
    if bad:
        raise RuntimeError("ticket publish order check failed (composite.hip):\n" + "\n".join(bad))
    if verbose:
        print("[isa] composite.hip: ticket publish order ok (partial store -> s_waitcnt vmcnt(0) -> ticket atomic)")


def build(force=False, verbose=True):
    os.makedirs(OBJ, exist_ok=True)
    os.makedirs(os.path.dirname(LIB), exist_ok=True)
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hpp")]
    headers.append(os.path.join(ROOT, "include", "cnerf.h"))
    jobs = []
    for s in sources():
        src, obj = os.path.join(CSRC, s), os.path.join(OBJ, s[:-4] + ".o")
        if force or _stale(obj, [src] + headers):
            jobs.append((src, obj))

    def cc(job):
        src, obj = job
        cmd = [HIPCC] + FLAGS + ["-c", src, "-o", obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        return src, r.returncode, r.stdout + r.stderr

    with ThreadPoolExecutor(max_workers=min(8, max(1, len(jobs)))) as ex:
        for src, rc, out in ex.map(cc, jobs):
            if verbose:
                print(f"[hipcc] {os.path.basename(src)} rc={rc}")
            if rc != 0:
                raise RuntimeError(f"hipcc failed on {src}:\n{out}")
    if any(os.path.basename(src) == "composite.hip" for src, _ in jobs):
        _check_ticket_isa(verbose)
    objs = [os.path.join(OBJ, s[:-4] + ".o") for s in sources()]
    if force or jobs or _stale(LIB, objs):
        r = subprocess.run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs,
                           capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("link failed:\n" + r.stdout + r.stderr)
        if verbose:
            print(f"[link] {LIB}")
    return LIB


if __name__ == "__main__":
    build(force="-f" in sys.argv)
