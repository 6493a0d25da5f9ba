"""dev tools: repository root on sys.path and the fixture sequence as the default data set."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ.setdefault("VS_DATASET_DIR", os.path.join(ROOT, "tests", "golden", "icl_nuim"))
