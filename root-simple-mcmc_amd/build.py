"""Build the HIP library in-tree: root-simple-mcmc_amd/lib/libsmcmc_amd.so.

hipcc cross-compiles for gfx950 without a GPU.  One object per translation unit;
the step kernels are split into one unit per (register-array size, likelihood)
so the units build in parallel.  Objects are rebuilt only when a source, a
header or the flags changed.
"""
import concurrent.futures
import hashlib
import os
import re
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
INCLUDE = os.path.join(ROOT, "include")
OBJ_DIR = os.path.join(HERE, "build")
LIB_DIR = os.path.join(HERE, "lib")
LIB_PATH = os.path.join(LIB_DIR, "libsmcmc_amd.so")

HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
ARCH = "gfx950"
# -ffp-contract=off: the kernels spell out every fused multiply-add; anything else
# must stay un-fused to match the reference's plain IEEE arithmetic bit for bit.
FLAGS = ["-std=c++17", "-O3", "-fPIC", "-ffp-contract=off", f"--offload-arch={ARCH}",
         "-fno-gpu-rdc", "-Wall", "-Wno-unused-function", f"-I{INCLUDE}", f"-I{CSRC}"]
LIKELIHOODS = (0, 1, 2)
STRESS_LIKELIHOODS = (4, 5, 6)   # SMCMC_LIKE_ASYM / HORRIFIC / CONSTRAINED: built for two families only (launch_step)
STRESS_DP = (31, 63)


def dp_list():
    text = open(os.path.join(CSRC, "smcmc_kernels.hip.h")).read()
    m = re.search(r"#define SMCMC_FOR_EACH_DP\(X\)(.*)", text)
    return [int(v) for v in re.findall(r"X\((\d+)\)", m.group(1))]


def _headers():
    hs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".h", ".hpp"))]
    hs += [os.path.join(INCLUDE, f) for f in os.listdir(INCLUDE)]
    return sorted(hs)


def _units(user_flag=None):
    """(source, defines, object name).  With a user likelihood only the two engines and the SMCMC_LIKE_USER
    instances are compiled with it; every other object is shared with the plain build."""
    engine = ("smcmc_engine.hip", [user_flag], "engine_user") if user_flag else ("smcmc_engine.hip", [], "engine")
    vaat = ("smcmc_vaat_engine.hip", [user_flag], "vaat_engine_user") if user_flag else ("smcmc_vaat_engine.hip", [], "vaat_engine")
    hmc = ("smcmc_hmc_engine.hip", [user_flag], "hmc_engine_user") if user_flag else ("smcmc_hmc_engine.hip", [], "hmc_engine")
    wave = (("smcmc_perchain_wave_inst.hip", [user_flag], "perchain_wave_user") if user_flag
            else ("smcmc_perchain_wave_inst.hip", [], "perchain_wave"))
    units = [engine, ("smcmc_selftest.hip", [], "selftest"), ("smcmc_autocorr.hip", [], "autocorr"),
             hmc, ("smcmc_hmc_mfma_inst.hip", [], "hmc_mfma"),
             vaat, ("smcmc_vaat_large.hip", [], "vaat_large"),
             ("smcmc_pooled_update.hip", [], "pooled_update"), ("smcmc_perchain_inst.hip", [], "perchain"),
             wave,
             ("smcmc_panel_mfma_inst.hip", [], "panel_mfma"), ("smcmc_fold_inst.hip", [], "fold")]
    for dp in dp_list():
        for like in LIKELIHOODS:
            units.append(("smcmc_inst.hip", [f"-DSMCMC_DP={dp}", f"-DSMCMC_LIKE={like}"], f"inst_dp{dp}_l{like}"))
    for dp in STRESS_DP:
        for like in STRESS_LIKELIHOODS:
            units.append(("smcmc_inst.hip", [f"-DSMCMC_DP={dp}", f"-DSMCMC_LIKE={like}"], f"inst_dp{dp}_l{like}"))
    for w in (4, 8):
        units.append(("smcmc_panel_inst.hip", [f"-DSMCMC_PANEL_W={w}"], f"panel_w{w}"))
        if user_flag:   # a user likelihood as an HMC target (finite-difference / covariant gradient)
            units.append(("smcmc_hmc_inst.hip", [f"-DSMCMC_PANEL_W={w}", user_flag], f"hmc_w{w}_user"))
        else:
            units.append(("smcmc_hmc_inst.hip", [f"-DSMCMC_PANEL_W={w}"], f"hmc_w{w}"))
    if user_flag:
        for dp in dp_list():   # SMCMC_LIKE_USER = 3
            units.append(("smcmc_inst.hip", [f"-DSMCMC_DP={dp}", "-DSMCMC_LIKE=3", user_flag], f"inst_dp{dp}_l3"))
        for w in (4, 8):       # 63 < dim <= 512 (served when the header defines SMCMC_USER_LIKELIHOOD_ANY_DIM)
            units.append(("smcmc_user_large.hip", [f"-DSMCMC_PANEL_W={w}", user_flag], f"user_large_w{w}"))
    return units


_EXTRA = {"files": []}   # the user likelihood header, part of the stamps of the units built with it
_STALE = []              # SMCMC_BUILD_ONLY: units linked although their sources changed


def _closure(path, seen):
    """`path` and the project headers it includes, transitively (quoted includes found in csrc/ or include/)."""
    if path in seen:
        return
    seen.add(path)
    for name in re.findall(r'^\s*#\s*include\s+"([^"]+)"', open(path).read(), flags=re.M):
        for base in (os.path.dirname(path), CSRC, INCLUDE):
            cand = os.path.join(base, name)
            if os.path.exists(cand):
                _closure(os.path.abspath(cand), seen)
                break


def _stamp(src, defs):
    h = hashlib.sha256()
    h.update(" ".join(FLAGS + defs).encode())
    extra = _EXTRA["files"] if any("SMCMC_USER_LIKELIHOOD" in d for d in defs) else []
    deps = set()
    _closure(os.path.abspath(os.path.join(CSRC, src)), deps)
    for path in extra:   # what the user's header includes (quoted, next to it or in csrc/ include/) counts too
        _closure(path, deps)
    for path in sorted(deps):
        h.update(os.path.basename(path).encode())
        h.update(open(path, "rb").read())
    return h.hexdigest()


def _compile(unit):
    src, defs, name = unit
    obj = os.path.join(OBJ_DIR, name + ".o")
    stamp_file = obj + ".stamp"
    stamp = _stamp(src, defs)
    if os.path.exists(obj) and os.path.exists(stamp_file) and open(stamp_file).read() == stamp:
        return obj, False
    # development aid: SMCMC_BUILD_ONLY=engine,inst_dp50_l0 recompiles just those units and links the rest as they
    # are (stale objects keep their old stamp and are rebuilt by the next full build)
    only = os.environ.get("SMCMC_BUILD_ONLY")
    if only and name not in only.split(",") and os.path.exists(obj):
        _STALE.append(name)
        return obj, False
    if src in CHECKED_SOURCES and _has_assembly_reads(defs):
        _compile_checked(src, defs, name, obj)
    else:
        cmd = [HIPCC] + FLAGS + defs + ["-c", os.path.join(CSRC, src), "-o", obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed for {name}:\n{' '.join(cmd)}\n{r.stdout}\n{r.stderr}")
    with open(stamp_file, "w") as f:
        f.write(stamp)
    return obj, True


# Translation units whose kernels place LDS reads and their waits by hand (inline assembly): compiled with the
# intermediate files kept, and the device listing checked before the object is accepted (inflight_check.py: the
# compiler must not touch a register while an assembly read into it is in flight).
CHECKED_SOURCES = ("smcmc_inst.hip",)


class InflightError(RuntimeError):
    """A unit whose listing failed the in-flight register check (every such unit of a build is reported together)."""


def _has_assembly_reads(defs):
    """kAsmReads<DP> of smcmc_kernels.hip.h: the families from 47 dimensions up place their LDS reads by hand."""
    for d in defs:
        m = re.match(r"-DSMCMC_DP=(\d+)$", d)
        if m:
            return int(m.group(1)) >= 47
    return True


def _compile_checked(src, defs, name, obj):
    import glob
    import shutil
    if HERE not in sys.path:
        sys.path.insert(0, HERE)
    import inflight_check
    tmp = os.path.join(OBJ_DIR, "tmp_" + name)
    shutil.rmtree(tmp, ignore_errors=True)
    os.makedirs(tmp)
    try:
        tmp_obj = os.path.join(tmp, name + ".o")
        cmd = [HIPCC] + FLAGS + defs + ["-save-temps=obj", "-c", os.path.join(CSRC, src), "-o", tmp_obj]
        r = subprocess.run(cmd, capture_output=True, text=True, cwd=tmp)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed for {name}:\n{' '.join(cmd)}\n{r.stdout}\n{r.stderr}")
        listings = glob.glob(os.path.join(tmp, f"*amdgcn*{ARCH}*.s"))
        if len(listings) != 1:
            raise RuntimeError(f"{name}: expected one device listing from -save-temps, found {listings}")
        report = []
        for kernel, reads, findings in inflight_check.check_listing(listings[0]):
            for idx, text, regs in findings[:6]:
                report.append(f"  {kernel} +{idx}: {text}   (assembly read in flight into v{regs})")
        if report:
            raise InflightError(f"{name}: the compiler touches registers with a hand-placed LDS read in flight "
                                f"(see inflight_check.py):\n" + "\n".join(report[:40]))
        os.replace(tmp_obj, obj)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def _link(objs, lib_path):
    cmd = [HIPCC, "-shared", "-fPIC", f"--offload-arch={ARCH}", "-fno-gpu-rdc"] + objs + ["-o", lib_path]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")


def build(jobs=None, verbose=False, user_likelihood=None, output=None, with_plain=False):
    """Compile every HIP translation unit for gfx950 and link the shared library.

    user_likelihood: a header defining smcmc_user_loglike<DP> (see smcmc_kernels.hip.h and
    examples/user_likelihood_asym.hip.h); the library built with it (default
    lib/libsmcmc_amd_user.so, or `output`) additionally serves SMCMC_LIKE_USER.  with_plain: also
    (re)build lib/libsmcmc_amd.so in the same pass, all units in one pool."""
    user_flag, user_lib = None, None
    if user_likelihood:
        user_likelihood = os.path.abspath(user_likelihood)
        user_flag = f'-DSMCMC_USER_LIKELIHOOD="{user_likelihood}"'
        _EXTRA["files"] = [user_likelihood]
        user_lib = output or os.path.join(LIB_DIR, "libsmcmc_amd_user.so")
    else:
        _EXTRA["files"] = []
    os.makedirs(OBJ_DIR, exist_ok=True)
    os.makedirs(LIB_DIR, exist_ok=True)
    plain_units = _units(None)
    user_units = _units(user_flag) if user_flag else []
    want_plain = with_plain or not user_flag
    todo, seen = [], set()
    for u in (plain_units if want_plain else []) + user_units:
        if u[2] not in seen:
            seen.add(u[2])
            todo.append(u)
    jobs = jobs or min(8, os.cpu_count() or 1)
    built = {}
    refused = []
    with concurrent.futures.ThreadPoolExecutor(max_workers=jobs) as pool:
        futures = [(unit, pool.submit(_compile, unit)) for unit in todo]
        for unit, fut in futures:
            try:
                obj, did = fut.result()
            except InflightError as exc:
                refused.append(str(exc))
                continue
            built[unit[2]] = (obj, did)
            if verbose and did:
                print("built", os.path.basename(obj), flush=True)
    if refused:
        raise InflightError("\n".join(refused))
    if _STALE:
        # a development shortcut, never a release build: objects of different source states may disagree about shared
        # structs (StepParams, PanelParams, ...)
        print("WARNING: SMCMC_BUILD_ONLY linked %d stale object(s) whose sources changed: %s\n"
              "         run a full build before trusting this library" % (len(_STALE), ", ".join(sorted(set(_STALE)))),
              file=sys.stderr, flush=True)
        del _STALE[:]
    result = None
    for units, lib_path in ((plain_units if want_plain else None, LIB_PATH), (user_units or None, user_lib)):
        if not units:
            continue
        objs = [built[u[2]][0] for u in units]
        if any(built[u[2]][1] for u in units) or not os.path.exists(lib_path):
            _link(objs, lib_path)
        result = lib_path
    return result


FROZEN_DEFINITION_FLAGS = ["-DSMCMC_PHILOX_ROUNDS=10", "-DSMCMC_NORMAL_TEXTBOOK=1"]
FROZEN_DEFINITION_UNITS = (("smcmc_inst.hip", ["-DSMCMC_DP=7", "-DSMCMC_LIKE=0"], "inst_dp7_l0"),
                           ("smcmc_inst.hip", ["-DSMCMC_DP=15", "-DSMCMC_LIKE=0"], "inst_dp15_l0"))
FROZEN_LIB_PATH = os.path.join(LIB_DIR, "libsmcmc_amd_frozen_definition.so")


def build_frozen_definition():
    """TEST LIBRARY of the frozen-definition golden set (tests/golden/frozen_definition_*.npz, include/smcmc_detmath.h):
    the plain library with the D <= 15 README-form step kernels recompiled with ten Philox rounds and the textbook normal
    pair.  Every other kernel in it still draws as the product does -- it serves tests/test_gpu_parity.py's frozen
    cases (D <= 15, SMCMC_LIKE_ISO_GAUSS, frozen / pooled covariance) and nothing else.  Call after build()."""
    objs = {u[2]: os.path.join(OBJ_DIR, u[2] + ".o") for u in _units(None)}
    rebuilt = False
    for src, defs, name in FROZEN_DEFINITION_UNITS:
        obj, did = _compile((src, defs + FROZEN_DEFINITION_FLAGS, name + "_frozen_definition"))
        objs[name] = obj
        rebuilt = rebuilt or did
    missing = [o for o in objs.values() if not os.path.exists(o)]
    if missing:
        raise RuntimeError("build_frozen_definition: run build() first (missing %s)" % ", ".join(map(os.path.basename, missing)))
    newest = max(os.path.getmtime(o) for o in objs.values())
    if rebuilt or not os.path.exists(FROZEN_LIB_PATH) or os.path.getmtime(FROZEN_LIB_PATH) < newest:
        _link(list(objs.values()), FROZEN_LIB_PATH)
    return FROZEN_LIB_PATH


if __name__ == "__main__":
    import argparse
    ap = argparse.ArgumentParser(description=__doc__)
    ap.add_argument("--user-likelihood", help="header defining smcmc_user_loglike<DP>: build a library that serves SMCMC_LIKE_USER")
    ap.add_argument("--output", help="path of the library built with --user-likelihood")
    a = ap.parse_args()
    print(build(verbose=True, user_likelihood=a.user_likelihood, output=a.output))
