"""Stage the reference's own Python front-end of lib/psa as a test fixture the GPU box can receive.

/root/reference does not exist on the GPU box, so `test_unmodified_reference_function_on_gpu` could never run where a
GPU exists.  This script (build container only) packs the three files the front-end consists of, byte for byte, into
tests/golden/ref_psa_frontend.npz (uint8 arrays) and prints their sha256; tests/test_psa_binding.py carries the same
digests, unpacks the files into a temporary package at test time and - here, where the reference is present -
additionally compares them with the files under /root/reference.  The fixture is test data (an input of the parity
test), not product source: nothing under lib/, model/ or semseg_amd/ reads it.

    python tests/golden/make_golden_refpsa.py
"""
import hashlib
import os

import numpy as np

REF = "/root/reference/lib/psa"
FILES = {"functional": "functional.py", "functions_init": "functions/__init__.py", "functions_psamask": "functions/psamask.py"}
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "ref_psa_frontend.npz")

if __name__ == "__main__":
    blobs = {}
    for key, rel in FILES.items():
        raw = open(os.path.join(REF, rel), "rb").read()
        blobs[key] = np.frombuffer(raw, dtype=np.uint8)
        print('    "%s": "%s",   # %s, %d bytes' % (key, hashlib.sha256(raw).hexdigest(), rel, len(raw)))
    np.savez_compressed(OUT, **blobs)
    print("wrote", OUT)
