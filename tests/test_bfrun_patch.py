"""CPU, build container only: the one host change the drop-in needs is a REAL patch
(patches/bfrun-bfhip.diff, SURVEY 8b "the one host patch", bfrun.c:1493-2008).  It is applied
here to a temporary copy of the reference's bfrun.c and the result is compiled with
`gcc -fsyntax-only -Wall` against the reference's own headers and include/bfhip.h -- with and
without -DBF_HAVE_BFHIP.  Nothing from the reference enters the repository or travels to the
GPU box (where /root/reference does not exist and this test is skipped)."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
PATCH = os.path.join(ROOT, "patches", "bfrun-bfhip.diff")

pytestmark = pytest.mark.skipif(not os.path.exists(os.path.join(REF, "bfrun.c")),
                                reason="the reference tree only exists in the build container")


def _apply(tmp_path):
    work = tmp_path / "host"
    work.mkdir()
    shutil.copy(os.path.join(REF, "bfrun.c"), work / "bfrun.c")
    r = subprocess.run(["patch", "-p1", "--no-backup-if-mismatch", "-i", PATCH], cwd=work,
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "FAILED" not in r.stdout and "fuzz" not in r.stdout, r.stdout
    return work


def _syntax(work, defines):
    return subprocess.run(["gcc", "-fsyntax-only", "-Wall", "-Werror=implicit-function-declaration",
                           "-Werror=incompatible-pointer-types", "-Werror=int-conversion"] + defines +
                          ["-I" + REF, "-I" + os.path.join(ROOT, "include"), "bfrun.c"],
                          cwd=work, capture_output=True, text=True)


def test_patch_applies_and_compiles_against_the_reference_headers(tmp_path):
    work = _apply(tmp_path)
    on = _syntax(work, ["-DBF_HAVE_BFHIP"])
    assert on.returncode == 0, on.stderr[-3000:]
    assert "warning" not in on.stderr, on.stderr[-3000:]
    off = _syntax(work, [])                              # without the define nothing changes
    assert off.returncode == 0, off.stderr[-3000:]
    src = (work / "bfrun.c").read_text()
    # the fused call sits between the two timestamps the survey names, once
    body = src[src.index("\ttimestamp(&t3);"):src.index("\ttimestamp(&t4);")]
    assert body.count("bfhip_period(") == 1 and "goto bfhip_period_done;" in body
    # n_processes > 1 is no longer refused: every forked filter process drives its own engine on
    # its own GPU, as a shard of the whole configuration
    assert "bfconf->n_processes != 1" not in src
    assert "process_index % n_devices" in src and "bfhip_engine_set_filter_active(" in src
    assert "BFHIP_COEFF_LAZY" in src and "bfhip_engine_set_output_active(" in src
    # `benchmark: true`: the GPU path adds device times into t[0..6] before the table is printed
    # (bfrun.c:2035-2078), and the reference's I/O delay (one period per call) is the default
    fused = body[body.index("bfhip_period("):body.index("goto bfhip_period_done;")]
    assert "bfhip_engine_stage_times(" in fused and "t[i] +=" in fused
    assert 'getenv("BFHIP_TWO_PERIODS")' in src and 'getenv("BFHIP_SYNC")' not in src
    # every bfhip_* function the patch calls is declared in include/bfhip.h
    import re
    called = set(re.findall(r"\b(bfhip_(?:engine|coeff)_[a-z_]+)\s*\(", src))
    header = open(os.path.join(ROOT, "include", "bfhip.h")).read()
    assert called and all(c in header for c in called), sorted(c for c in called if c not in header)


def test_committed_patch_is_what_the_generator_produces(tmp_path):
    before = open(PATCH).read()
    r = subprocess.run(["python3", os.path.join(ROOT, "tools", "make_bfrun_patch.py"), REF],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert open(PATCH).read() == before


def test_every_convolver_symbol_the_host_objects_need_is_exported(tmp_path):
    """the drop-in claim at link level: compile the reference's host files that reach the convolver
    (the patched bfrun.c, bfconf.c, delay.c, dai.c, dither.c, firwindow.c) to objects in a temporary
    directory and check that every `convolver_*` / `bfhip_*` symbol they leave undefined is a
    dynamic export of libbfhip.so -- the library stands where fftw_convolver.o + convolver_xmm.o
    stood (reference Makefile:43-44).  (The whole binary cannot be linked here: the configuration
    lexer needs flex.)"""
    work = _apply(tmp_path)
    objs = []
    for name, extra in (("bfrun.c", ["-DBF_HAVE_BFHIP", "-I" + os.path.join(ROOT, "include")]),
                        ("bfconf.c", []), ("delay.c", []), ("dai.c", []), ("dither.c", []), ("firwindow.c", [])):
        src = str(work / name) if name == "bfrun.c" else os.path.join(REF, name)
        obj = str(tmp_path / (name[:-2] + ".o"))
        r = subprocess.run(["gcc", "-c", "-O0", "-w", "-I" + REF] + extra + [src, "-o", obj],
                           capture_output=True, text=True)
        assert r.returncode == 0, name + ": " + r.stderr[-1500:]
        objs.append(obj)
    undef = subprocess.run(["nm", "-u"] + objs, capture_output=True, text=True).stdout.split()
    need = sorted({s for s in undef if s.startswith(("convolver_", "bfhip_"))})
    assert len(need) >= 40, need                        # the 21 convolver.h symbols the host uses + the fused API
    import brutefir_amd as bf
    dyn = subprocess.run(["nm", "-D", "--defined-only", bf.LIB_PATH], capture_output=True, text=True).stdout.split()
    missing = [s for s in need if s not in dyn]
    assert not missing, missing


def test_reference_filter_process_binary_is_built_without_placeholders():
    """oracle/_ref/ref_filter_process -- the reference's own filter_process() (bfrun.c unchanged) over the
    product's convolver.h symbols, tests/test_gpu_refloop.py's checker -- is (re)built by `make -C oracle
    ref` from the sources where they lie, linked with -z defs; every undefined symbol it has left is
    libc's, libm's or libbfhip.so's"""
    import brutefir_amd as bf
    assert os.path.exists(bf.LIB_PATH), "build the product first (python -m brutefir_amd.build)"
    r = subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "ref"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    exe = os.path.join(ROOT, "oracle", "_ref", "ref_filter_process")
    assert os.path.exists(exe)
    undef = subprocess.run(["nm", "-D", "--undefined-only", exe], capture_output=True, text=True).stdout.split()
    ours = sorted(s for s in undef if s.startswith(("convolver_", "bfhip_")))
    assert len(ours) >= 15 and all(s.startswith("convolver_") for s in ours), ours     # the convolver.h boundary, nothing else
    dyn = subprocess.run(["nm", "-D", "--defined-only", bf.LIB_PATH], capture_output=True, text=True).stdout.split()
    assert all(s in dyn for s in ours)
    # the same harness over bfrun.c WITH the host patch applied: the fused entry points come in, the
    # temporary patched copy is gone after the build
    exe2 = os.path.join(ROOT, "oracle", "_ref", "ref_filter_process_bfhip")
    assert os.path.exists(exe2)
    undef2 = subprocess.run(["nm", "-D", "--undefined-only", exe2], capture_output=True, text=True).stdout.split()
    fused = sorted(s for s in undef2 if s.startswith("bfhip_"))
    assert "bfhip_engine_rt_block" in fused and "bfhip_engine_set_filter_active" in fused and "bfhip_engine_stage_times" in fused
    assert all(s in dyn for s in fused)
    assert not os.path.exists(os.path.join(ROOT, "oracle", "_ref", "patched"))
    recipe = open(os.path.join(ROOT, "oracle", "Makefile")).read()
    assert "defsym" not in recipe.split("$(FPROC):")[1].split("endif")[0]


def test_the_fused_period_keeps_one_barrier_between_the_filter_processes():
    """all filter processes wake on one pipe (bfrun.c:833, 2488: n_processes tokens a period); the fused
    path must keep a synch_filter_processes() in front of its period or a fast process takes two tokens
    (tests/test_gpu_refloop.py::test_filter_processes_keep_step_on_their_shared_wake_pipe shows it on the GPU)"""
    diff = open(os.path.join(ROOT, "patches", "bfrun-bfhip.diff")).read()
    added = [ln[1:].strip() for ln in diff.splitlines() if ln.startswith("+") and not ln.startswith("+++")]
    i_bar = added.index("synch_filter_processes(filter_readfd, filter_writefd, process_index);")
    i_per = next(i for i, ln in enumerate(added) if ln.startswith("bfhip_period(icomm_fctrl"))
    assert i_bar < i_per and i_per - i_bar <= 2
