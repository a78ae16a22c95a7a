"""tools/ hygiene (no GPU): every developer script parses, and every one is named in tools/README.md with what it measures —
a script nobody can find the purpose of is deleted, not kept."""
import ast
import glob
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_every_tool_parses_and_is_documented():
    readme = open(os.path.join(ROOT, "tools", "README.md")).read()
    scripts = sorted(glob.glob(os.path.join(ROOT, "tools", "*.py")) + glob.glob(os.path.join(ROOT, "tools", "*.sh")) +
                     glob.glob(os.path.join(ROOT, "tools", "micro", "*.hip")))
    scripts = [s for s in scripts if not os.path.basename(s).startswith("_")]          # _g*.sh / _diag*.py: scratch job wrappers, git-ignored
    assert len(scripts) > 20
    for s in scripts:
        name = os.path.basename(s)
        assert name in readme or os.path.splitext(name)[0] in readme, f"tools/{name} is not described in tools/README.md"
        if s.endswith(".py"):
            ast.parse(open(s).read(), filename=s)
