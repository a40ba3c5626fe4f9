#!/usr/bin/env python3
"""Undefined-name check for the package's modules without a linter in the image: every Name loaded in a module must be bound
somewhere in it (import, def, class, assignment, argument, comprehension / loop / with / except target) or be a builtin.
    python tools/check_names.py dipole_normal_prop_amd/*.py"""
import ast
import builtins
import sys

bad = 0
for path in sys.argv[1:]:
    tree = ast.parse(open(path).read(), path)
    bound = set(dir(builtins)) | {"__file__", "__name__", "__doc__"}
    for node in ast.walk(tree):
        if isinstance(node, (ast.Import, ast.ImportFrom)):
            for a in node.names:
                bound.add((a.asname or a.name).split(".")[0])
        elif isinstance(node, (ast.FunctionDef, ast.AsyncFunctionDef, ast.ClassDef)):
            bound.add(node.name)
            if not isinstance(node, ast.ClassDef):
                for a in node.args.args + node.args.kwonlyargs + node.args.posonlyargs:
                    bound.add(a.arg)
                for a in (node.args.vararg, node.args.kwarg):
                    if a:
                        bound.add(a.arg)
        elif isinstance(node, ast.Lambda):
            for a in node.args.args + node.args.kwonlyargs:
                bound.add(a.arg)
            for a in (node.args.vararg, node.args.kwarg):
                if a:
                    bound.add(a.arg)
        elif isinstance(node, ast.Name) and isinstance(node.ctx, (ast.Store, ast.Del)):
            bound.add(node.id)
        elif isinstance(node, ast.ExceptHandler) and node.name:
            bound.add(node.name)
        elif isinstance(node, (ast.Global, ast.Nonlocal)):
            bound.update(node.names)
    for node in ast.walk(tree):
        if isinstance(node, ast.Name) and isinstance(node.ctx, ast.Load) and node.id not in bound:
            print(f"{path}:{node.lineno}: undefined name {node.id}")
            bad += 1
sys.exit(1 if bad else 0)
