"""tools/variants/*.patch hold experiment code that is not in the product tree (tools/README.md).  A patch that no
longer applies to the tree it documents is dead weight: keep them applicable."""
import glob
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("patch", sorted(glob.glob(os.path.join(ROOT, "tools", "variants", "*.patch"))), ids=os.path.basename)
def test_variant_patch_applies_to_the_tree(patch):
    if shutil.which("git") is None or not os.path.isdir(os.path.join(ROOT, ".git")):
        pytest.skip("not a git checkout")
    r = subprocess.run(["git", "apply", "--check", patch], cwd=ROOT, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
