"""The campaign / session tools under tools/ are not part of the product, but the evidence under profiles/ was made with them: they must
at least stay valid Python (compiled, not run -- they need a GPU), and every campaign summary must name a tool that exists."""
import glob
import os
import py_compile
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_tools_compile():
    files = sorted(glob.glob(os.path.join(ROOT, "tools", "*.py")) + glob.glob(os.path.join(ROOT, "tools", "r6", "*.py")))
    assert len(files) >= 20
    for f in files:
        py_compile.compile(f, doraise=True)


def test_campaign_summaries_name_their_tools():
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "r6_lab", "fuzz_*.txt"))):
        text = open(f).read()
        named = set(re.findall(r"tools/r6/(fuzz_[a-z_]+\.py)", text))
        if os.path.basename(f) == "fuzz_final_tree.txt":
            continue
        assert named, f
        for t in named:
            assert os.path.exists(os.path.join(ROOT, "tools", "r6", t)), (f, t)
