"""Source hygiene for the native code (a trailing `//` once commented out the `WS(crossV, ...)` that shared its 300-character
line, and the encoder epilogue then wrote through a null pointer).  Three guards: bounded line length, no statement hidden behind a comment on
the same line, and a workspace table with one buffer per line whose name is the field it fills."""
import glob
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SOURCES = sorted(glob.glob(os.path.join(ROOT, "streamkit_amd", "csrc", "*.hip")) + glob.glob(os.path.join(ROOT, "streamkit_amd", "csrc", "*.h")) +
                 glob.glob(os.path.join(ROOT, "streamkit_amd", "csrc", "*.cpp")) + glob.glob(os.path.join(ROOT, "include", "*.h")) +
                 glob.glob(os.path.join(ROOT, "oracle", "*.[ch]")))
MAX_LINE = 200


def _comment_start(line):
    """index of a // comment outside string / char literals, or -1"""
    in_s, i = None, 0
    while i < len(line):
        ch = line[i]
        if in_s:
            if ch == "\\":
                i += 2
                continue
            if ch == in_s:
                in_s = None
        elif ch in "\"'":
            in_s = ch
        elif line.startswith("//", i):
            return i
        i += 1
    return -1


def test_line_length_is_bounded():
    """<= 200 columns; the only exemption is a line that is one string literal (the plugins' JSON schemas, one property per line)"""
    bad = []
    for p in SOURCES:
        for n, line in enumerate(open(p, encoding="utf-8"), 1):
            line = line.rstrip("\n")
            if len(line) > MAX_LINE and not line.lstrip().startswith('"'):
                bad.append("%s:%d (%d columns)" % (os.path.relpath(p, ROOT), n, len(line)))
    assert not bad, bad


def test_no_statement_hides_behind_a_comment():
    """what follows `//` on a line must not look like code that allocates, checks or launches: WS( / want( / HIPCHK( / hipMalloc / hipLaunchKernelGGL / a trailing `;`
    after a call.  (A comment may of course MENTION these names; what is refused is the statement form `name(...);`.)"""
    stmt = re.compile(r"\b(WS|want|HIPCHK|hipMalloc|hipHostMalloc|hipMemcpyAsync|hipLaunchKernelGGL|hipExtLaunchKernelGGL|ws_alloc)\s*(<[^>]*>)?\s*\(.*\)\s*;")
    bad = []
    for p in SOURCES:
        for n, line in enumerate(open(p, encoding="utf-8"), 1):
            k = _comment_start(line)
            if k >= 0 and stmt.search(line[k:]):
                bad.append("%s:%d: %s" % (os.path.relpath(p, ROOT), n, line[k:k + 100].strip()))
    assert not bad, bad


def test_workspace_table_is_one_buffer_per_line():
    src = open(os.path.join(ROOT, "streamkit_amd", "csrc", "skw_engine.hip"), encoding="utf-8").read()
    body = src[src.index("auto want = [&]"):src.index("const char* ws_failed = nullptr;")]
    entries = [l for l in body.split("\n") if "want(" in l and "auto want" not in l]
    assert len(entries) >= 45
    names = []
    for l in entries:
        m = re.fullmatch(r"\s*want\(\"(\w+)\", c->(\w+), .*, (true|false)\);", l)
        assert m, "not a single table entry on its own line: %r" % l
        assert m.group(1) == m.group(2), "table name %s fills field %s" % (m.group(1), m.group(2))
        assert l.count("want(") == 1
        names.append(m.group(1))
    assert len(set(names)) == len(names)
    # the buffers the encoder / decoder kernels are handed must all be in the table
    for need in ("crossK", "crossV", "selfK", "selfV", "logits", "x", "y16", "Qh", "Kh", "Vt", "hbuf", "dx", "dy16", "dq16", "datt16", "dh16", "st", "toks", "static_mask"):
        assert need in names, need
    # and skw_ctx_create / the entry points verify the table by name
    assert "ws_first_null(c)" in src and src.count("WS_READY(c);") >= 5


def test_one_switchboard_one_getenv():
    """the product used to ship ~50 `getenv` switches, most of them measured-and-lost experiments no parity test had seen.  Now: ONE table
    (skw_engine.hip g_sw_defs), ONE getenv in the engine library (the table's loader) plus the reference's own KOKORO_EXECUTION_PROVIDER (config.rs:55-58), and every row of
    the table has a case in tests/test_gpu_switches.py.  No GPU needed: the table is read through the library's debug entry points."""
    import importlib.util
    import sys
    calls = []
    for p in SOURCES:
        if os.sep + "oracle" + os.sep in p:
            continue
        for n, line in enumerate(open(p, encoding="utf-8").read().split("\n"), 1):
            c = _comment_start(line)
            code = line if c < 0 else line[:c]
            if "getenv(" in code:
                calls.append((os.path.basename(p), n, code.strip()))
    assert len(calls) == 2, calls
    assert any(f == "skw_engine.hip" and "g_sw_defs[i].name" in code for f, _, code in calls) and any("KOKORO_EXECUTION_PROVIDER" in code for _, _, code in calls), calls
    sys.path.insert(0, ROOT)
    from streamkit_amd import engine
    table = engine.switches()
    assert all(cur == dflt for dflt, cur, _ in table.values()) or any(k.startswith("SKW_") for k in os.environ)
    spec = importlib.util.spec_from_file_location("_sw_cases", os.path.join(ROOT, "tests", "test_gpu_switches.py"))
    src = open(spec.origin).read()
    names = set(re.findall(r'^\s*\("([A-Z0-9_]+)", -?\d+, "(?:same_bits|tolerance|quant|resample)"\)', src, flags=re.M)) | set(re.findall(r'\("([A-Z0-9_]+)", -?\d+, "(?:same_bits|tolerance|quant|resample)"\)', src))
    assert names == set(table), (sorted(names), sorted(table))
    readme = open(os.path.join(ROOT, "README.md")).read()
    for name in table:
        assert "SKW_" + name in readme, "README.md does not document SKW_" + name


def test_only_test_infrastructure_touches_the_oracle():
    """The oracle is the checker: only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import, call, link or execute anything under oracle/.  The product
    (streamkit_amd/: Python host side, C++ / HIP sources, the Makefile) and tools/ never do — they may NAME it in comments; they may not load it."""
    import re
    load = re.compile(r"oracle_lib|libskw_oracle|Oracle(Model|Tts|Silero|Resampler|Decoder)|dlopen\([^)]*oracle|-lskw_oracle|#include\s+\"[^\"]*oracle")
    offenders = []
    for base in ("streamkit_amd", "tools"):
        for dp, _, fs in os.walk(os.path.join(ROOT, base)):
            if os.sep + "build" in dp or "__pycache__" in dp:
                continue
            for f in fs:
                if f.endswith((".py", ".hip", ".cpp", ".h", ".c", ".sh")) or f == "Makefile":
                    p = os.path.join(dp, f)
                    for n, line in enumerate(open(p, encoding="utf-8", errors="replace"), 1):
                        if load.search(line) and not line.lstrip().startswith(("//", "#", "*", "/*")):
                            offenders.append("%s:%d" % (os.path.relpath(p, ROOT), n))
    assert not offenders, offenders
    # bench.py: every load of the oracle sits inside the cpu_baseline leg (after its marker comment, before the function that holds it ends)
    src = open(os.path.join(ROOT, "bench.py"), encoding="utf-8").read().split("\n")
    hits = [i for i, line in enumerate(src) if load.search(line) and not line.lstrip().startswith("#")]
    marker = next(i for i, line in enumerate(src) if "CPU baseline: the oracle" in line)
    assert hits and all(marker < i < marker + 40 for i in hits), (hits, marker)
    entry = open(os.path.join(ROOT, "__graft_entry__.py"), encoding="utf-8").read()
    assert entry.index("def smoke") < entry.index("oracle_lib")                              # (build() only compiles the checker)


def test_every_python_tool_and_hunt_script_compiles():
    """tools/ and tests/hunt/ are run by hand on the GPU box, rarely: a syntax error there would only show up in the middle of a measurement."""
    bad = []
    files = [os.path.join(ROOT, f) for f in ("bench.py", "__graft_entry__.py", os.path.join("tests", "pin_against_whisper_cpp.py"))]
    for base in ("tools", os.path.join("tests", "hunt"), "streamkit_amd"):
        for dp, _, fs in os.walk(os.path.join(ROOT, base)):
            if "__pycache__" not in dp:
                files += [os.path.join(dp, f) for f in fs if f.endswith(".py")]
    for f in files:
        try:
            compile(open(f, encoding="utf-8").read(), f, "exec")
        except SyntaxError as e:
            bad.append("%s:%s: %s" % (os.path.relpath(f, ROOT), e.lineno, e.msg))
    assert len(files) > 30 and not bad, bad
