#!/usr/bin/env python3
"""Extract golden constants from the reference's shipped TF MetaGraphDef files.

Run in the BUILD container only (needs /root/reference); writes small data
fixtures next to this script.  No TensorFlow needed: the protobuf wire format is
decoded by hand (MetaGraphDef.graph_def=2 -> GraphDef.node=1 ->
NodeDef{name=1, op=2, input=3, device=4, attr=5(map: key=1,value=2)};
AttrValue{list=1,s=2,i=3,f=4,b=5,type=6,shape=7,tensor=8};
TensorProto{dtype=1,tensor_shape=2,tensor_content=4,float_val=5,int_val=7}).

The fixtures are DATA the reference produced (initial particle grids, scalar
constants, variable shapes, op order of the hot path); they are the only
reference-produced golden vectors for this path.
"""
import json
import os
import struct
import sys

import numpy as np

REF = "/root/reference"
META = {
    "walk": "ckpt_DeepMimicWalk-v0/deepmimic_dppo_pfpn_particle35/34114/model.ckpt-78000.meta",
    "kick": "ckpt_DeepMimicKick-v0/deepmimic_dppo_pfpn_particle35/34114/model.ckpt-78000.meta",
    "punch": "ckpt_DeepMimicPunch-v0/deepmimic_dppo_pfpn_particle35/34114/model.ckpt-78128.meta",
}


def varint(buf, i):
    r = 0
    s = 0
    while True:
        b = buf[i]
        i += 1
        r |= (b & 0x7F) << s
        if not b & 0x80:
            return r, i
        s += 7


def fields(buf):
    """Yield (field_number, wire_type, value) for one message."""
    i = 0
    n = len(buf)
    while i < n:
        key, i = varint(buf, i)
        fn, wt = key >> 3, key & 7
        if wt == 0:
            v, i = varint(buf, i)
        elif wt == 1:
            v = buf[i:i + 8]
            i += 8
        elif wt == 2:
            ln, i = varint(buf, i)
            v = buf[i:i + ln]
            i += ln
        elif wt == 5:
            v = buf[i:i + 4]
            i += 4
        else:
            raise ValueError("wire type %d" % wt)
        yield fn, wt, v


def parse_shape(buf):
    dims = []
    for fn, wt, v in fields(buf):
        if fn == 2:  # Dim
            size = 0
            for f2, w2, v2 in fields(v):
                if f2 == 1:
                    size = v2 if v2 < (1 << 63) else v2 - (1 << 64)
            dims.append(size)
    return dims


def parse_tensor(buf):
    dtype, shape, content, fvals, ivals = None, [], None, [], []
    for fn, wt, v in fields(buf):
        if fn == 1:
            dtype = v
        elif fn == 2:
            shape = parse_shape(v)
        elif fn == 4:
            content = bytes(v)
        elif fn == 5:
            if wt == 2:
                fvals.extend(struct.unpack("<%df" % (len(v) // 4), v))
            else:
                fvals.append(struct.unpack("<f", v)[0])
        elif fn in (7, 10, 11):      # int_val / int64_val / bool_val
            if wt == 2:
                j = 0
                while j < len(v):
                    x, j = varint(v, j)
                    ivals.append(x if x < (1 << 63) else x - (1 << 64))
            else:
                ivals.append(v if v < (1 << 63) else v - (1 << 64))
        elif fn == 6:                # double_val
            if wt == 2:
                fvals.extend(struct.unpack("<%dd" % (len(v) // 8), v))
            else:
                fvals.append(struct.unpack("<d", v)[0])
    np_dt = {1: np.float32, 3: np.int32, 9: np.int64, 2: np.float64, 10: np.bool_}.get(dtype)
    if np_dt is None:
        return None
    n = int(np.prod(shape)) if shape else 1
    if content is not None:
        arr = np.frombuffer(content, dtype=np_dt).copy()
    elif fvals:
        arr = np.asarray(fvals, dtype=np_dt)
        if arr.size == 1 and n > 1:
            arr = np.full(n, arr[0], dtype=np_dt)
    elif ivals:
        arr = np.asarray(ivals, dtype=np_dt)
        if arr.size == 1 and n > 1:
            arr = np.full(n, arr[0], dtype=np_dt)
    else:
        arr = np.zeros(n, dtype=np_dt)
    return arr.reshape(shape)


def parse_attr(buf):
    out = {}
    for fn, wt, v in fields(buf):
        if fn == 8:
            out["tensor"] = parse_tensor(v)
        elif fn == 7:
            out["shape"] = parse_shape(v)
        elif fn == 3:
            out["i"] = v if v < (1 << 63) else v - (1 << 64)
        elif fn == 4:
            out["f"] = struct.unpack("<f", v)[0]
        elif fn == 2:
            out["s"] = bytes(v).decode("utf8", "replace")
        elif fn == 5:
            out["b"] = bool(v)
        elif fn == 6:
            out["type"] = v
    return out


def parse_node(buf):
    node = {"name": None, "op": None, "input": [], "device": "", "attr": {}}
    for fn, wt, v in fields(buf):
        if fn == 1:
            node["name"] = bytes(v).decode()
        elif fn == 2:
            node["op"] = bytes(v).decode()
        elif fn == 3:
            node["input"].append(bytes(v).decode())
        elif fn == 4:
            node["device"] = bytes(v).decode()
        elif fn == 5:
            k, val = None, None
            for f2, w2, v2 in fields(v):
                if f2 == 1:
                    k = bytes(v2).decode()
                elif f2 == 2:
                    val = parse_attr(v2)
            node["attr"][k] = val
    return node


def load_nodes(path):
    buf = memoryview(open(path, "rb").read())
    nodes = []
    for fn, wt, v in fields(buf):
        if fn == 2:  # graph_def
            for f2, w2, v2 in fields(v):
                if f2 == 1:
                    nodes.append(parse_node(v2))
    return nodes


def main():
    here = os.path.dirname(os.path.abspath(__file__))
    summary = {}
    for tag, rel in META.items():
        nodes = load_nodes(os.path.join(REF, rel))
        by_name = {n["name"]: n for n in nodes}
        info = {"n_nodes": len(nodes)}
        # variables of global_net: name -> shape
        variables = {}
        for n in nodes:
            if n["op"] in ("VariableV2", "Variable") and n["name"].startswith("global_net/") \
                    and "/Adam" not in n["name"]:
                variables[n["name"]] = n["attr"].get("shape", {}).get("shape")
        info["variables"] = variables
        placeholders = {n["name"]: n["attr"].get("shape", {}).get("shape")
                        for n in nodes if n["op"] == "Placeholder" and n["name"].startswith("global_net/")}
        info["placeholders"] = placeholders
        # scalar float constants inside global_net + optimizer (name -> value)
        scalars = {}
        for n in nodes:
            if n["op"] == "Const" and n["name"].split("/")[0] in (
                    "global_net", "optimizer", "cond", "Adam", "beta1_power", "beta2_power", "GreaterEqual"):
                t = n["attr"].get("value", {}).get("tensor")
                if t is not None and t.size == 1 and t.dtype in (np.float32, np.float64):
                    scalars[n["name"]] = float(t.reshape(-1)[0])
        info["scalar_consts"] = scalars
        # particle grid initialisers
        for vn in ("global_net/actor/samples", "global_net/actor/samples_std"):
            init = by_name.get(vn + "/Initializer/Const")
            arr = init["attr"]["value"]["tensor"]
            assert arr.dtype == np.float32
            if tag == "walk":
                np.save(os.path.join(here, "meta_" + vn.split("/")[-1] + "_init.npy"), arr)
            info[vn + ":init_shape"] = list(arr.shape)
            info[vn + ":init_crc"] = int(np.frombuffer(arr.tobytes(), dtype=np.uint32).sum() & 0xFFFFFFFF)
        # op order of the hot-path forward (names only, in graph order), for docs/tests
        hot = [(n["name"], n["op"]) for n in nodes
               if n["name"].startswith("global_net/actor/") and "summary" not in n["name"]
               and n["op"] not in ("Const", "Identity", "VariableV2", "Assign", "Shape", "StridedSlice", "Pack")]
        info["actor_ops"] = hot[:400]
        rs = [(n["name"], n["op"]) for n in nodes if n["name"].startswith("global_net/resample/")
              and n["op"] not in ("Const", "Identity", "Shape", "StridedSlice", "Pack")]
        info["resample_ops"] = rs[:600]
        info["cond_ops"] = [(n["name"], n["op"]) for n in nodes if n["name"].startswith("cond/")
                            and n["op"] not in ("Const", "Identity", "Shape", "StridedSlice", "Pack", "Switch", "Merge")]
        info["logprob_ops"] = [(n["name"], n["op"], n["input"]) for n in nodes
                               if n["name"].startswith(("global_net/policy_loss/pi_target/",
                                                        "global_net/actor/Normal/prob_1/"))
                               and n["op"] not in ("Const", "Shape", "StridedSlice", "Pack")]
        info["multinomial"] = [
            {"name": n["name"], "seed": n["attr"].get("seed", {}).get("i"), "seed2": n["attr"].get("seed2", {}).get("i"),
             "output_dtype": n["attr"].get("output_dtype", {}).get("type")}
            for n in nodes if n["op"] == "Multinomial" and n["name"].startswith("global_net/")]
        summary[tag] = info
    with open(os.path.join(here, "meta_constants.json"), "w") as f:
        json.dump(summary, f, indent=1, sort_keys=True)
    print("wrote", os.path.join(here, "meta_constants.json"))


if __name__ == "__main__":
    sys.exit(main())
