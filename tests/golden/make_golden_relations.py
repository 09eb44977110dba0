#!/usr/bin/env python3
"""Golden for dataio.read_relations: the first 40 lines of the reference's DataSet/RawData/intel.relations (data) and the JSON
the reference's own DataPreprocess/preprocess_relation.py writes for them.  The script is RUN (in a scratch directory laid
out the way its hard-coded relative paths expect), never copied.  Development container only: python tests/golden/make_golden_relations.py"""
import os
import shutil
import subprocess
import sys
import tempfile

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
with open(os.path.join(REF, "DataSet", "RawData", "intel.relations")) as f:
    head = f.readlines()[:40]
head.append(head[3])                      # a repeated pair of stamps: the last line wins in the reference's dicts
with tempfile.TemporaryDirectory() as tmp:
    os.makedirs(os.path.join(tmp, "DataSet"))
    with open(os.path.join(tmp, "DataSet", "intel.relations"), "w") as f:
        f.writelines(head)
    subprocess.run([sys.executable, "-B", os.path.join(REF, "DataPreprocess", "preprocess_relation.py")], cwd=tmp, check=True)
    shutil.copy(os.path.join(tmp, "DataSet", "intel.relations"), os.path.join(HERE, "relations_excerpt.txt"))
    shutil.copy(os.path.join(tmp, "DataSet", "intel_relation_processed"), os.path.join(HERE, "relations_excerpt_processed.json"))
print("written")
