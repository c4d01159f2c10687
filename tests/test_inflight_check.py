"""root-simple-mcmc_amd/inflight_check.py: the listing check build.py applies to every step-kernel unit (no compiler
instruction may touch a register while a hand-placed LDS read into it is in flight)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "root-simple-mcmc_amd"))
import inflight_check  # noqa: E402

HEAD = "_ZN5smcmc11step_kernelILi31ELi0ELb1ELb0ELb1ELb0EEEvNS_10StepParamsE:\n"
TAIL = ".Lfunc_end0:\n"


def _check(body, tmp_path):
    path = tmp_path / "k.s"
    path.write_text(HEAD + body + TAIL)
    (name, reads, findings), = inflight_check.check_listing(str(path))
    assert name.startswith("_ZN5smcmc11step_kernel")
    return reads, findings


def test_spill_behind_the_read_is_found(tmp_path):
    # what the compiler did in the 31-dimension family: the prefetched operand spilled before its data was there
    reads, findings = _check("""
\t;;#ASMSTART
\tds_read_b64 v[8:9], v12 offset:32
\t;;#ASMEND
\tv_accvgpr_write_b32 a57, v9
\tv_accvgpr_write_b32 a56, v8
\tv_mul_f64 v[0:1], v[2:3], v[4:5]
""", tmp_path)
    assert reads == 1 and [f[2] for f in findings] == [[9], [8]]


def test_waits_cover_reads_in_issue_order(tmp_path):
    # two reads; a wait that leaves one in flight completes the older one only
    reads, findings = _check("""
\t;;#ASMSTART
\tds_read_b128 v[20:23], v1 offset:16
\t;;#ASMEND
\t;;#ASMSTART
\tds_read_b128 v[24:27], v1 offset:32
\t;;#ASMEND
\t;;#ASMSTART
\ts_waitcnt lgkmcnt(1)
\t;;#ASMEND
\tv_mul_f64 v[0:1], v[20:21], v[22:23]
\tv_mul_f64 v[2:3], v[24:25], v[4:5]
\ts_waitcnt lgkmcnt(0)
\tv_mul_f64 v[2:3], v[26:27], v[4:5]
""", tmp_path)
    assert reads == 2 and len(findings) == 1 and findings[0][2] == [24, 25]


def test_the_compilers_own_lds_operations_count(tmp_path):
    # a compiler read behind the assembly read is one more operation in the queue: lgkmcnt(1) then completes the assembly read
    reads, findings = _check("""
\t;;#ASMSTART
\tds_read_b64 v[8:9], v12
\t;;#ASMEND
\tds_read_b64 v[30:31], v13
\ts_waitcnt lgkmcnt(1)
\tv_add_f64 v[0:1], v[8:9], v[2:3]
""", tmp_path)
    assert reads == 1 and findings == []


def test_writes_into_a_register_in_flight_are_found_too(tmp_path):
    reads, findings = _check("""
\t;;#ASMSTART
\tds_read_b64 v[12:13], v12 offset:64
\t;;#ASMEND
\tv_accvgpr_read_b32 v12, a102
""", tmp_path)
    assert len(findings) == 1


def test_build_checks_exactly_the_families_with_assembly_reads():
    """build.py decides from -DSMCMC_DP which units to check; the kernel header decides which families place their LDS
    reads by hand (kAsmReads<DP>).  The two thresholds are one number."""
    import re
    import importlib.util
    spec = importlib.util.spec_from_file_location("smcmc_build", os.path.join(ROOT, "root-simple-mcmc_amd", "build.py"))
    build = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(build)
    header = open(os.path.join(ROOT, "root-simple-mcmc_amd", "csrc", "smcmc_kernels.hip.h")).read()
    threshold = int(re.search(r"constexpr bool kAsmReads = DP >= (\d+);", header).group(1))
    for dp in build.dp_list():
        assert build._has_assembly_reads([f"-DSMCMC_DP={dp}", "-DSMCMC_LIKE=0"]) == (dp >= threshold), dp
    assert "smcmc_inst.hip" in build.CHECKED_SOURCES
