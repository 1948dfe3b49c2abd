"""Every entry point declared in include/*.h must say which reference code it replaces (a file:line citation
into fp/ or Source/) or state that it has no reference counterpart."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CITE = re.compile(r"(fp/[A-Za-z]+\.(cpp|hpp):\d+|Source/PluginProcessor\.(cpp|h):\d+|:\d+-\d+|no reference counterpart)", re.I)


def _functions_with_comments(path):
    """(function name, the block comment that documents it): block comments start a line; comments inside a
    parameter list or trailing a declaration do not document the declarations that follow."""
    src = open(path).read()
    out, comment, pos = [], "", 0
    for m in re.finditer(r"/\*.*?\*/", src, flags=re.S):
        for name in re.findall(r"\b(irb_[a-z_0-9]+)\s*\(", src[pos:m.start()]):
            out.append((name, comment))
        line_start = src.rfind("\n", 0, m.start()) + 1
        if src[line_start:m.start()].strip() == "":
            comment = m.group(0)
        pos = m.end()
    for name in re.findall(r"\b(irb_[a-z_0-9]+)\s*\(", src[pos:]):
        out.append((name, comment))
    return out


def test_every_declaration_cites_the_reference_or_says_it_has_no_counterpart():
    missing = []
    for h in ("irb_fdl.h", "irb_deconv.h"):
        for name, comment in _functions_with_comments(os.path.join(ROOT, "include", h)):
            if not CITE.search(comment):
                missing.append(f"{h}:{name}")
    assert not missing, "declarations without a reference citation: " + ", ".join(missing)
