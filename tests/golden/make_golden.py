#!/usr/bin/env python3
"""Regenerate tests/golden/ref_components_{fast,ieee}.json.gz.

Runs ONLY in the build container (needs /root/reference).  It builds the component-level
reference harness (oracle/Makefile target `ref`: the reference's own sources compiled where they
lie, our driver oracle/ref_harness/ref_components.cc) in two flavours —

  fast : -O3 -ffast-math -DFAST_MATH -DFAST_TRIG   (the reference's release flags, CMakeLists.txt:241)
  ieee : -O2 -ffp-contract=off -DFAST_MATH -DFAST_TRIG

— runs both and stores their stdout (inputs + outputs as IEEE-754 bit patterns) as fixtures.
The fixtures are data; no reference source text is stored.
"""
import gzip
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))


def main():
    if not os.path.isdir("/root/reference"):
        sys.exit("reference tree not present; fixtures can only be regenerated in the build container")
    subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "ref"], check=True, timeout=900)
    which = sys.argv[1:] or ["components", "textures", "integrator"]
    for name in which:          # textures: image textures + shader nodes, oracle/ref_harness/ref_textures.cc
        for variant in ("fast", "ieee"):
            exe = os.path.join(ROOT, "oracle", "_ref", f"ref_{name}_{variant}")
            out = subprocess.run([exe], check=True, capture_output=True, timeout=120).stdout
            out = out[out.index(b"{\n"):]          # the reference's handlers log to stdout before the document starts
            path = os.path.join(HERE, f"ref_{name}_{variant}.json.gz")
            with gzip.GzipFile(path, "wb", mtime=0) as f:
                f.write(out)
            print(path, len(out), "bytes raw")


if __name__ == "__main__":
    main()
