"""Build-container-only: the example and evaluation drivers are this build's own programs.  Each function of
examples/*.py and examples/evaluation/*.py that has a namesake in the reference's file of the same name is tokenised
(comments and string literals dropped) and compared with that namesake; the similarity ratio must stay below 0.35
(the round-3 drivers were re-wrapped copies at 0.7-0.99).  Skipped where /root/reference does not exist
(the GPU box): nothing of the reference travels."""
import ast
import difflib
import io
import os
import tokenize

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
FILES = ["examples/example_pandas_Jointspace.py", "examples/example_pandas_cartesian.py",
         "examples/example_pointmasses_static.py", "examples/example_pointmasses_dynamic.py",
         "examples/evaluation/evaluate_horizon.py", "examples/evaluation/evaluate_random_dynamic_scenarios.py"]
LIMIT = 0.35

pytestmark = pytest.mark.skipif(not os.path.isdir(REF), reason="reference tree not present")


def _functions(path):
    """name -> (source of the whole function, source of its body without the signature)."""
    src = open(path).read()
    lines = src.split("\n")
    out = {}
    for node in ast.walk(ast.parse(src)):
        if isinstance(node, (ast.FunctionDef, ast.AsyncFunctionDef)):
            out[node.name] = ("\n".join(lines[node.lineno - 1:node.end_lineno]),
                              "\n".join(lines[node.body[0].lineno - 1:node.end_lineno]))
    return out, src


def _tokens(text):
    toks = []
    try:
        for t in tokenize.generate_tokens(io.StringIO(text).readline):
            if t.type in (tokenize.COMMENT, tokenize.STRING, tokenize.NL, tokenize.NEWLINE, tokenize.INDENT, tokenize.DEDENT,
                          tokenize.ENDMARKER):
                continue
            toks.append(t.string)
    except (tokenize.TokenError, IndentationError):
        pass
    return toks


def _ratio(a, b):
    return difflib.SequenceMatcher(None, _tokens(a), _tokens(b), autojunk=False).ratio()


@pytest.mark.parametrize("rel", FILES)
def test_driver_functions_are_not_the_references(rel):
    """Bodies are compared without their signatures: names, positional parameters and defaults are the contract
    (tests/test_examples_contract.py) and identical by design.  The whole file, signatures included, must pass too."""
    mine, mine_src = _functions(os.path.join(ROOT, rel))
    theirs, theirs_src = _functions(os.path.join(REF, rel))
    shared = sorted(set(mine) & set(theirs))
    assert shared, "the contract functions must exist under the reference's names"
    report = {name: round(_ratio(mine[name][1], theirs[name][1]), 3) for name in shared}
    assert all(r < LIMIT for r in report.values()), report
    with_signatures = {name: round(_ratio(mine[name][0], theirs[name][0]), 3) for name in shared}
    assert all(r < 0.5 for r in with_signatures.values()), with_signatures
    whole = round(_ratio(mine_src, theirs_src), 3)
    assert whole < LIMIT, (whole, report)
