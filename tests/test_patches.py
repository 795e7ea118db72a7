"""tools/patches holds work that was built and inspected but not yet run on the device: each patch
must keep applying to the tree it was written against, or it is dead text."""
import glob
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(shutil.which("git") is None or not os.path.isdir(os.path.join(ROOT, ".git")),
                    reason="needs the git checkout")
@pytest.mark.parametrize("patch", sorted(glob.glob(os.path.join(ROOT, "tools", "patches", "*.patch"))))
def test_patch_still_applies_or_is_already_applied(patch):
    forward = subprocess.run(["git", "apply", "--check", patch], cwd=ROOT, capture_output=True)
    if forward.returncode == 0:
        return
    # (applied and verified in a later round: then it reverses cleanly and should be deleted)
    backward = subprocess.run(["git", "apply", "--check", "-R", patch], cwd=ROOT,
                              capture_output=True)
    assert backward.returncode == 0, forward.stderr.decode()
