"""ASan + UBSan over the library's host-only code (blob codec, ring ops, trainers, parameter
parsing), fed with the reference's golden triples, lifted rows and iris triples.  CPU build only:
GPU sanitizers are not available on the pool."""
import os
import struct
import subprocess

import numpy as np

from oracle import oracle
from triple_fmt import dict_to_blob

HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "sanitize")


def test_host_code_is_clean_under_asan_and_ubsan(goldens, tmp_path):
    subprocess.check_call(["make", "-s", "-C", HERE, "host_sanitize"])
    blobs = []
    for fname, doc in goldens.items():
        for t in doc["tests"]:
            for e in t["expected"]:
                blobs.append(np.asarray(dict_to_blob(e["value"]), dtype=np.float64))
    rng = np.random.default_rng(4)
    num = [rng.normal(size=200).astype(np.float32) for _ in range(3)]
    cat = [rng.integers(0, 4, 200).astype(np.int32) * 7 - 3 for _ in range(3)]
    blobs.append(oracle.State(oracle.WIDE).update(num, cat).finalize())
    blobs.append(oracle.State(oracle.WIDE).update(num, cat, nb=True).finalize())
    blobs.append(oracle.State(oracle.WIDE).update(num, []).finalize())
    blobs.append(oracle.State(oracle.WIDE).update([], cat).finalize())
    blobs += oracle.lift([c[:3] for c in num], [c[:3] for c in cat])
    path = tmp_path / "blobs.bin"
    with open(path, "wb") as fh:
        fh.write(struct.pack("<Q", len(blobs)))
        for b in blobs:
            b = np.ascontiguousarray(b, dtype=np.float64)
            fh.write(struct.pack("<Q", b.size))
            fh.write(b.tobytes())
    out = subprocess.run([os.path.join(HERE, "host_sanitize"), str(path)], capture_output=True, text=True,
                         timeout=600, env=dict(os.environ, ASAN_OPTIONS="detect_leaks=1"))
    assert out.returncode == 0, out.stdout + out.stderr
    assert "host_sanitize ok" in out.stdout
