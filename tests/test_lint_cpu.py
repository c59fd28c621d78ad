"""Static guards over the Python that only ever executes on a GPU box (tests/*_gpu.py helpers, scripts/, the
cuda branches of the drivers).  The CPU suite runs those files on the host emulator, where `_dev()` is `cpu`, so a
bug on the `cuda` branch is invisible here -- round 4 lost its hardware verdict to a helper that called itself
(`_sync()` inside `_sync()`).  Two checks, no GPU:

  * AST: no function calls itself by its own bare name unless it is listed in RECURSIVE_OK, and every
    global name a function body loads resolves to a module-level binding, an import or a builtin;
  * behaviour: every zero-argument module-level helper of a `tests/*_gpu.py` file whose name says "sync" is CALLED
    with the device faked to `cuda` and `torch.cuda.synchronize` counted.
"""
import ast
import builtins
import glob
import importlib
import os
import symtable

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FILES = sorted(glob.glob(os.path.join(ROOT, "tests", "*.py")) + glob.glob(os.path.join(ROOT, "tests", "tools", "*.py"))
               + glob.glob(os.path.join(ROOT, "scripts", "*.py")) + glob.glob(os.path.join(ROOT, "permafrost-engine_amd", "*.py"))
               + glob.glob(os.path.join(ROOT, "oracle", "*.py")) + glob.glob(os.path.join(ROOT, "oracle", "ref", "*.py"))
               + [os.path.join(ROOT, "bench.py"), os.path.join(ROOT, "__graft_entry__.py")])
RECURSIVE_OK = {("test_lint_cpu.py", "_unresolved"), ("bench.py", "r4")}          # (file basename, function name) pairs that recurse on purpose


def _rel(p):
    return os.path.relpath(p, ROOT)


def _self_calls(tree):
    """(function name, line) of every call of a function to its own bare name from inside its own body (nested
    definitions of the same name excluded)."""
    found = []

    class V(ast.NodeVisitor):
        def __init__(self):
            self.stack = []

        def visit_FunctionDef(self, node):
            self.stack.append(node.name)
            self.generic_visit(node)
            self.stack.pop()

        visit_AsyncFunctionDef = visit_FunctionDef

        def visit_ClassDef(self, node):
            # a method named f calling a module-level f() is not recursion
            self.stack.append(None)
            self.generic_visit(node)
            self.stack.pop()

        def visit_Call(self, node):
            if self.stack and isinstance(node.func, ast.Name) and self.stack[-1] == node.func.id:
                # only a true self call when the enclosing function is not a method
                if len(self.stack) < 2 or self.stack[-2] is not None:
                    found.append((node.func.id, node.lineno))
            self.generic_visit(node)

    V().visit(tree)
    return found


@pytest.mark.parametrize("path", FILES, ids=_rel)
def test_no_function_calls_itself(path):
    tree = ast.parse(open(path).read(), path)
    bad = [(n, ln) for n, ln in _self_calls(tree) if (os.path.basename(path), n) not in RECURSIVE_OK]
    assert not bad, "%s: self-recursive call(s) %s -- a search-and-replace accident?" % (_rel(path), bad)


def _module_bindings(tree):
    names = set(dir(builtins)) | {"__file__", "__name__", "__doc__", "__builtins__", "__spec__", "__package__"}
    for node in ast.walk(tree):
        if isinstance(node, (ast.FunctionDef, ast.AsyncFunctionDef, ast.ClassDef)):
            names.add(node.name)
        elif isinstance(node, ast.Import):
            names.update((a.asname or a.name).split(".")[0] for a in node.names)
        elif isinstance(node, ast.ImportFrom):
            names.update(a.asname or a.name for a in node.names)
        elif isinstance(node, ast.Name) and isinstance(node.ctx, (ast.Store, ast.Del)):
            names.add(node.id)
        elif isinstance(node, ast.Global):
            names.update(node.names)
        elif isinstance(node, ast.ExceptHandler) and node.name:
            names.add(node.name)
    return names


def _unresolved(table, known, out, path):
    for child in table.get_children():
        if child.get_type() == "function":
            for sym in child.get_symbols():
                if sym.is_global() and sym.is_referenced() and not sym.is_assigned() and sym.get_name() not in known:
                    out.append((child.get_name(), child.get_lineno(), sym.get_name()))
        _unresolved(child, known, out, path)


@pytest.mark.parametrize("path", FILES, ids=_rel)
def test_every_global_name_resolves(path):
    src = open(path).read()
    if "import *" in src:
        pytest.skip("star import")
    known = _module_bindings(ast.parse(src, path))
    out = []
    _unresolved(symtable.symtable(src, path, "exec"), known, out, path)
    assert not out, "%s: names that resolve to nothing at run time: %s" % (_rel(path), out)


GPU_TEST_MODULES = sorted(os.path.basename(p)[:-3] for p in glob.glob(os.path.join(ROOT, "tests", "*_gpu.py")))


@pytest.mark.parametrize("modname", GPU_TEST_MODULES)
def test_gpu_sync_helpers_run_on_a_faked_cuda_device(modname, monkeypatch):
    """The `cuda` branch of every `*sync*` helper, executed: device faked, torch.cuda.synchronize counted."""
    import torch
    monkeypatch.delenv("NAVHIP_LIB", raising=False)
    mod = importlib.import_module("tests." + modname)
    helpers = [n for n, f in vars(mod).items()
               if "sync" in n.lower() and callable(f) and getattr(f, "__module__", None) == mod.__name__
               and not n.startswith("test") and f.__code__.co_argcount == 0]
    if not helpers:
        pytest.skip("no zero-argument sync helper in tests/%s.py" % modname)
    calls = []
    monkeypatch.setattr(torch.cuda, "synchronize", lambda *a, **k: calls.append(1))
    if hasattr(mod, "_dev"):
        monkeypatch.setattr(mod, "_dev", lambda: torch.device("cuda", 0))
    for n in helpers:
        before = len(calls)
        getattr(mod, n)()
        assert len(calls) == before + 1, "tests/%s.py::%s did not reach torch.cuda.synchronize" % (modname, n)
