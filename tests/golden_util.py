import json
import os

from oracle import kzg_model as M

HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    return json.load(open(os.path.join(HERE, name)))


def sc(h):
    return int.from_bytes(bytes.fromhex(h), "little")


def pt(h):
    return M.g1_from_compressed(bytes.fromhex(h))
