"shared by the summarise scripts: note in profiles/<tag>_tree.json which sources a collection was taken from"
import json
import os
import time


def record(tag, collection, src_dir):
    path = f'profiles/{tag}_tree.json'
    rec = json.load(open(path)) if os.path.exists(path) else {
        'what': 'sha256 of csr_amd/csrc + include/csrk.h (tools/tree_stamp.py) on the GPU box when each collection of this '
                'round\'s profiles ran; tests/test_profiles_fresh.py compares them with the tree', 'collections': {}}
    rec['collections'][collection] = {'csrc_sha256': open(os.path.join(src_dir, 'tree.txt')).read().strip(),
                                       'summarised': time.strftime('%Y-%m-%d %H:%M:%S')}
    json.dump(rec, open(path, 'w'), indent=1)
