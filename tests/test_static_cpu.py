"""A name that is used but never bound anywhere in its module (a forgotten ``import math`` in a branch that only runs when
an in-kernel wait expires during the start-up trials was found by a soak run, not by a test): every module of the package,
``bench.py`` and ``__graft_entry__.py`` are parsed and each loaded name must be bound somewhere in the same file or be a
builtin.  Cruder than a linter (scopes are not modelled), but it needs nothing that is not installed."""
import ast
import builtins
import glob
import os

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _unbound(path):
    tree = ast.parse(open(path).read())
    bound = set(dir(builtins)) | {"__file__", "__name__", "__doc__"}
    for n in ast.walk(tree):
        if isinstance(n, (ast.Import, ast.ImportFrom)):
            bound.update((a.asname or a.name).split(".")[0] for a in n.names)
        elif isinstance(n, (ast.FunctionDef, ast.AsyncFunctionDef, ast.ClassDef)):
            bound.add(n.name)
        elif isinstance(n, ast.Name) and isinstance(n.ctx, (ast.Store, ast.Del)):
            bound.add(n.id)
        elif isinstance(n, ast.ExceptHandler) and n.name:
            bound.add(n.name)
        elif isinstance(n, (ast.Global, ast.Nonlocal)):
            bound.update(n.names)
        if isinstance(n, (ast.FunctionDef, ast.AsyncFunctionDef, ast.Lambda)):
            a = n.args
            bound.update(x.arg for x in a.args + a.kwonlyargs + a.posonlyargs)
            bound.update(x.arg for x in (a.vararg, a.kwarg) if x is not None)
    return ["%s:%d %s" % (os.path.relpath(path, REPO), n.lineno, n.id) for n in ast.walk(tree)
            if isinstance(n, ast.Name) and isinstance(n.ctx, ast.Load) and n.id not in bound]


def test_no_name_is_used_without_being_bound_in_its_module():
    files = sorted(glob.glob(os.path.join(REPO, "distributed_tensorflow_amd", "**", "*.py"), recursive=True))
    files += [os.path.join(REPO, "bench.py"), os.path.join(REPO, "__graft_entry__.py")]
    files += sorted(glob.glob(os.path.join(REPO, "demo*", "*.py"))) + sorted(glob.glob(os.path.join(REPO, "retrain*", "*.py")))
    assert len(files) > 40
    bad = [m for f in files for m in _unbound(f)]
    assert not bad, bad
