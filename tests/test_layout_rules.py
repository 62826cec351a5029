"""CPU: the oracle is test infrastructure -- only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline() may import it; the
product package, the tools and the entry scripts never do, and the package has no CPU fallback to route through."""
import ast
import glob
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _oracle_imports(path):
    """(function name or None, line) of every `import oracle...` / `from oracle... import` in a file"""
    tree = ast.parse(open(path).read(), path)
    found = []

    def visit(node, fn):
        for child in ast.iter_child_nodes(node):
            name = child.name if isinstance(child, (ast.FunctionDef, ast.AsyncFunctionDef)) else fn
            if isinstance(child, ast.Import) and any(a.name.split(".")[0] == "oracle" for a in child.names):
                found.append((fn, child.lineno))
            if isinstance(child, ast.ImportFrom) and (child.module or "").split(".")[0] == "oracle" and child.level == 0:
                found.append((fn, child.lineno))
            visit(child, name)
    visit(tree, None)
    return found


def test_only_tests_smoke_and_the_cpu_baseline_import_the_oracle():
    for path in sorted(glob.glob(os.path.join(ROOT, "dynamic-rs-segmentation_amd", "*.py")) + glob.glob(os.path.join(ROOT, "tools", "*.py"))
                       + glob.glob(os.path.join(ROOT, "drs_amd", "*.py")) + glob.glob(os.path.join(ROOT, "*_dilated_random.py"))):
        assert _oracle_imports(path) == [], path
    assert {fn for fn, _ in _oracle_imports(os.path.join(ROOT, "bench.py"))} == {"cpu_baseline"}
    assert {fn for fn, _ in _oracle_imports(os.path.join(ROOT, "__graft_entry__.py"))} == {"smoke"}
    # and the oracle itself says what it is
    head = open(os.path.join(ROOT, "oracle", "__init__.py")).read().lower()
    assert "test infrastructure" in head and "parity unpinned" in head
