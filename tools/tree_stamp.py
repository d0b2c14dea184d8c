#!/usr/bin/env python3
"""
sha256 over the library's sources (csr_amd/csrc/*, include/csrk.h: names and contents, sorted) -- what a profile was taken
from.  The collection scripts write it beside their summaries (profiles/rNN_tree.json); tests/test_profiles_fresh.py
fails when the newest round's profiles were taken from other sources than the tree holds.  (The GPU box has no .git:
a content hash works there and here.)
    python tools/tree_stamp.py            -> prints the hash
"""
import glob
import hashlib
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def stamp(root=ROOT):
    h = hashlib.sha256()
    files = sorted(glob.glob(os.path.join(root, 'csr_amd', 'csrc', '*.hip')) + glob.glob(os.path.join(root, 'csr_amd', 'csrc', '*.h')) +
                   [os.path.join(root, 'include', 'csrk.h')])
    for f in files:
        h.update(os.path.relpath(f, root).encode() + b'\0')
        h.update(open(f, 'rb').read())
        h.update(b'\0')
    return h.hexdigest()


if __name__ == '__main__':
    print(stamp())
