#!/usr/bin/env python3
"""Copies two of the reference's shipped INPUT files into tests/golden (build container only: /root/reference does not
travel).  Data only -- no source text.

* k17hp.snap        the whole file (load_data(5), kernel.cu:1007-1011; 10 002 bodies, 1.2 MB), byte for byte.
* k17c.snap         the whole file (load_data(4), kernel.cu:1001-1005; 32 770 bodies, 3.9 MB), byte for byte.
* stars.dat         the whole file (load_data(3), kernel.cu:996-1000; 43 802 records "z y x vz vy vx", 3.2 MB), byte for byte.
* stars_8192.dat    its first 8192 records (a system small enough for the ten-frame CPU restatement): the number tokens are
                    the file's own, unchanged, six to a line.
"""
import hashlib
import os
import shutil

REF = "/root/reference/main_project/data"
HERE = os.path.dirname(os.path.abspath(__file__))


def sha(path):
    return hashlib.sha256(open(path, "rb").read()).hexdigest()


def main():
    shutil.copyfile(os.path.join(REF, "k17hp.snap"), os.path.join(HERE, "k17hp.snap"))
    shutil.copyfile(os.path.join(REF, "k17c.snap"), os.path.join(HERE, "k17c.snap"))
    shutil.copyfile(os.path.join(REF, "stars.dat"), os.path.join(HERE, "stars.dat"))
    tok = open(os.path.join(REF, "stars.dat")).read().split()
    assert len(tok) % 6 == 0 and len(tok) // 6 == 43802
    with open(os.path.join(HERE, "stars_8192.dat"), "w") as f:
        for r in range(8192):
            f.write(" " + " ".join(tok[6 * r:6 * r + 6]) + "\n")
    for name in ("k17hp.snap", "k17c.snap", "stars.dat", "stars_8192.dat"):
        print(name, os.path.getsize(os.path.join(HERE, name)), sha(os.path.join(HERE, name)))
    print("source k17hp.snap", sha(os.path.join(REF, "k17hp.snap")), "stars.dat", sha(os.path.join(REF, "stars.dat")))


if __name__ == "__main__":
    main()
