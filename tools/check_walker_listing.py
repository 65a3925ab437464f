"""Reads the gfx950 listing of the walker kernels and checks the shape of their node-phase loops.

    python tools/check_walker_listing.py            # compiles render.hip to listings (CPU only, 15 s) and checks them
    python tools/check_walker_listing.py file.s ... # checks listings made elsewhere

Why: the node phase (rt_traverse.h, node_phase) is a loop whose end test is the same in every lane of a wave
(two ballot counts), with one `if (lane may step)` inside.  hipcc builds it with execution masks, and has two ways
of doing so.  In one, all lanes stay in the loop until the common exit: the ballots see the whole wave and the lane
flag `popped` (a boolean, held as a scalar lane mask) is merged only under the mask of the lanes that stepped.  In
the other, lanes that do not step leave an inner loop and rejoin outside it: the ballots then run with those lanes
masked out, and the flag of a lane that leaves before the last trip is rebuilt from a register that holds nothing
for it -- a lane that waits right after a pop comes back with the flag cleared, uses a stack entry twice and ends up
indexing memory with a wrapped stack pointer.  Which form comes out depends on small things (the thresholds of the
end test did it once, a readfirstlane round the end test another time), so the listing is looked at before a
changed walker goes anywhere near a GPU.

The check: the loop header that holds the phase test (two s_bcnt1_i32_b64 within a few lines of each other, markers of inline
assembly aside) must not be the target of an
`s_cbranch_execnz` -- that branch is the inner loop's "some lanes stepped, go round again without the others".
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "raytracing_amd", "csrc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math", "-Wno-unused-function",
         "-Wno-unused-command-line-argument", "--cuda-device-only", "-S"]
# (source, extra flags): every translation unit that instantiates traverse_rays, as raytracing_amd/csrc/Makefile builds them
UNITS = [("render.hip", []), ("render.hip", ["-DTRV_STACK_LDS=2", "-DRT_STACK_DEPTH=8"])]


def functions(text):
    """(name, lines) of every function of a listing."""
    name, body = None, []
    for line in text.splitlines():
        m = re.match(r"^(_Z\w+):", line)
        if m and name is None:
            name, body = m.group(1), []
        elif name is not None:
            body.append(line)
            if line.startswith(".Lfunc_end"):
                yield name, body
                name = None


def check_function(lines):
    """-> (number of phase-test loops, [header labels built the wrong way])"""
    bad, loops = [], 0
    lines = [x for x in lines if not x.strip().startswith(";;#ASM")]     # the counts are made in inline assembly (lane_count)
    skip = -1
    for i in range(len(lines) - 1):
        second = [k for k in range(i + 1, min(i + 5, len(lines))) if "s_bcnt1_i32_b64" in lines[k]]
        if i > skip and "s_bcnt1_i32_b64" in lines[i] and second:      # the two counts, at most a few scalar instructions apart
            skip = second[0]
            header = None
            for j in range(i, max(i - 60, -1), -1):               # nearest loop header above the test
                m = re.match(r"^(\.LBB\d+_\d+):", lines[j])
                if m:
                    # the label's comment runs over the lines that follow it, one per enclosing loop
                    note, k = lines[j], j + 1
                    while k < len(lines) and lines[k].lstrip().startswith(";") and not re.match(r"^\S", lines[k]):
                        note += lines[k]
                        k += 1
                    if "Loop Header" in note:
                        header = m.group(1)
                        break
            if header is None:
                bad.append("phase test at line %d has no loop header above it" % i)
                continue
            loops += 1
            pat = re.compile(r"\bs_cbranch_execnz\s+" + re.escape(header) + r"\b")
            if any(pat.search(x) for x in lines):
                bad.append(header)
    return loops, bad


def check_listing(text):
    report = {}
    for name, body in functions(text):
        if "k_traverse" not in name or "packets" in name:
            continue
        report[name] = check_function(body)
    return report


def compile_listing(source, extra):
    with tempfile.TemporaryDirectory() as tmp:
        out = os.path.join(tmp, "listing.s")
        subprocess.run(["/opt/rocm/bin/hipcc"] + FLAGS + extra + [source, "-o", out], cwd=CSRC, check=True,
                       stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        with open(out) as f:
            return f.read()


def main(argv):
    texts = [(p, open(p).read()) for p in argv] if argv else [(" ".join([s] + e), compile_listing(s, e)) for s, e in UNITS]
    failed = False
    for label, text in texts:
        report = check_listing(text)
        if not report:
            print("%s: no walker kernels found" % label)
            failed = True
        for name, (loops, bad) in sorted(report.items()):
            print("%s: %s: %d node-phase loops, %s" % (label, name, loops, "ok" if not bad and loops else "WRONG SHAPE " + ", ".join(bad)))
            failed |= bool(bad) or loops == 0
    return 1 if failed else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
