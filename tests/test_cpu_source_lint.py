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
