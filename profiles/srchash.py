#!/usr/bin/env python3
"""Identity of the kernel sources a set of profile figures was collected from: sha256 over the files of
lightspinner_amd/csrc that define the device code and the launch plan (sorted by name).  profiles/summarize.py stores it
in pmc_figures.json; bench.py compares it with the sources beside the library it loads and labels the figures stale when
they differ; tests/test_profile_figures.py fails when the committed figures do not belong to the committed sources."""
import glob
import hashlib
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, 'lightspinner_amd', 'csrc')


def csrc_files():
    return sorted(f for pat in ('*.hip', '*.h', '*.cpp') for f in glob.glob(os.path.join(CSRC, pat)))


def csrc_hash():
    h = hashlib.sha256()
    for f in csrc_files():
        h.update(os.path.basename(f).encode() + b'\0')
        h.update(open(f, 'rb').read())
    return h.hexdigest()[:16]


if __name__ == '__main__':
    print(csrc_hash())
