"""
The committed profiles of the newest round (profiles/rNN_*) must have been taken from the library sources the tree holds:
every collection script stamps a hash of csr_amd/csrc + include/csrk.h into profiles/rNN_tree.json (tools/tree_stamp.py).
A kernel edited after its profile was taken makes this fail until the profiles are collected again.
"""
import glob
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_newest_profiles_match_the_sources():
    sys.path.insert(0, os.path.join(ROOT, 'tools'))
    from tree_stamp import stamp
    files = glob.glob(os.path.join(ROOT, 'profiles', 'r*_tree.json'))
    assert files, 'no profiles/rNN_tree.json: run tools/collect_profiles.sh and tools/summarise_profiles.py'
    newest = max(files, key=lambda f: int(re.search(r'r(\d+)_tree', f).group(1)))
    rec = json.load(open(newest))
    now = stamp()
    stale = {k: v for k, v in rec['collections'].items() if v['csrc_sha256'] != now}
    assert not stale, f'{os.path.basename(newest)}: taken from other sources than the tree holds: {sorted(stale)} (now {now[:12]})'
    for need in ('spmv', 'configs'):
        assert need in rec['collections'], f'{os.path.basename(newest)} has no {need} collection'
