"""CPU (hipcc cross-compiles without a GPU): the resident form of the tiled kernel keeps its barriers convergent.

The request loop of expand_tile_kernel<D, K, ONE, SVC = true> has barriers around sections that only wave 0 executes
(polling the doorbell, publishing `done`).  Written with `if (threadIdx.x == 0)`, hipcc threaded lanes 1..63 of wave 0
into the next trip's s_barrier while lane 0 still had its store to do, and the workgroup hung on the device
(profiles/micro/mailbox_latency.hip reproduces it).  The kernel therefore lets all 64 lanes of wave 0 act together; this
test pins what that buys in the generated code: every service instantiation has exactly the barriers of the source --
four of its own (two around the poll, two around the report), where the ordinary form has the one in front of its
completion word -- and its polling loops branch on scalar conditions (wave-uniform), not on exec masks."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "motion_primitive_library_amd", "csrc", "expand_tile_kernel.hip")


def _hipcc():
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    return None


@pytest.mark.skipif(_hipcc() is None, reason="hipcc not available")
def test_service_instantiations_have_the_barriers_of_the_source(tmp_path):
    out = tmp_path / "tile.s"
    cmd = [_hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-S", "--cuda-device-only",
           SRC, "-o", str(out)]
    subprocess.run(cmd, check=True, capture_output=True, cwd=os.path.dirname(SRC))
    text = out.read_text()
    # kernels: label "<mangled>:" ... "s_endpgm"
    kernels = {}
    for m in re.finditer(r"^(_ZN4mplx\S*expand_tile_kernelILi(\d)ELi(\d)ELb([01])ELb([01])E\S*):\s*;", text, re.M):
        body = text[m.end():text.index("s_endpgm", m.end())]
        kernels[(int(m.group(2)), int(m.group(3)), int(m.group(4)), int(m.group(5)))] = body
    assert len(kernels) == 32  # D 2/3 x K 1..4 x ONE x SVC
    for (d, k, one, svc), body in kernels.items():
        if not svc:
            continue
        plain = kernels[(d, k, one, 0)]
        nb, nb0 = body.count("s_barrier"), plain.count("s_barrier")
        # (3 or 4: hipcc folds the ordinary form's last barrier into the tile loop's in some instantiations)
        assert 3 <= nb - nb0 <= 4, "expand_tile_kernel<%d,%d,%d>: %d barriers in the service form, %d in the ordinary one" % (d, k, one, nb, nb0)
        # the doorbell poll: a system-scope load (sc0 sc1) inside a loop closed by a SCALAR branch
        poll = body.index("s_memrealtime")
        window = body[poll:poll + 4000]
        assert "sc0 sc1" in window and re.search(r"s_cbranch_(vccz|vccnz|scc0|scc1)", window), (d, k, one)
        assert "s_dcache_inv" in body and "buffer_wbl2 sc0 sc1" in body
        assert "s_memrealtime" not in plain and "s_dcache_inv" not in plain


@pytest.mark.skipif(_hipcc() is None, reason="hipcc not available")
def test_every_wave_drains_its_stores_before_the_completion_barrier(tmp_path):
    """Round-3 advice: `done` (service form) and the DoneSignal word (ordinary form) are published by wave 0 after a
    barrier, and the host reads the landing block as soon as the word appears.  On gfx9 the barrier's workgroup-scope
    release waits for lgkmcnt only, so every wave has to drain its own vmcnt first: each instantiation carries an explicit
    `s_waitcnt vmcnt(0)` (inline asm, kept verbatim by the compiler) whose next memory-or-barrier instruction is the
    s_barrier in front of the publication -- no store of the wave can slip in between."""
    out = tmp_path / "tile.s"
    cmd = [_hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-S", "--cuda-device-only",
           SRC, "-o", str(out)]
    subprocess.run(cmd, check=True, capture_output=True, cwd=os.path.dirname(SRC))
    text = out.read_text()
    drain = re.compile(r";;#ASMSTART\s*\n\s*s_waitcnt vmcnt\(0\)\s*\n\s*;;#ASMEND")
    n = 0
    for m in re.finditer(r"^(_ZN4mplx\S*expand_tile_kernelILi(\d)ELi(\d)ELb([01])ELb([01])E\S*):\s*;", text, re.M):
        body = text[m.end():re.compile(r"^\.Lfunc_end\d+:", re.M).search(text, m.end()).start()]  # (s_endpgm also ends early exits)
        hits = list(drain.finditer(body))
        assert len(hits) >= 1, m.group(1)
        for h in hits:
            nxt = re.search(r"^\s*(global_\w+|buffer_\w+|flat_\w+|s_barrier)", body[h.end():], re.M)
            assert nxt and nxt.group(1) == "s_barrier", (m.group(1), nxt.group(1) if nxt else None)
        n += 1
    assert n == 32
