#!/usr/bin/env python
"""Extracts the inference-relevant constants / op attributes from the reference's own
MetaGraphDef (model_2000000_qp30~35.dat.meta, written by TF 1.4.1) into
tests/golden/meta_constants.json -- run HERE (needs /root/reference), commit the JSON.

No TensorFlow / protobuf schema needed: a generic protobuf wire-format walk over
MetaGraphDef.graph_def.node[*] (name, op, input, attr).  The JSON is data (numbers, strides,
shapes, node wiring), not reference source.
"""
import json
import os
import struct
import sys

META = "/root/reference/HM-16.5_Test_AI/bin/model_2000000_qp30~35.dat.meta"
HERE = os.path.dirname(os.path.abspath(__file__))


def varint(b, i):
    v = s = 0
    while True:
        c = b[i]
        i += 1
        v |= (c & 0x7F) << s
        s += 7
        if not c & 0x80:
            return v, i


def fields(b):
    i = 0
    while i < len(b):
        tag, i = varint(b, i)
        f, w = tag >> 3, tag & 7
        if w == 0:
            v, i = varint(b, i)
        elif w == 1:
            v, i = b[i:i + 8], i + 8
        elif w == 2:
            n, i = varint(b, i)
            v, i = b[i:i + n], i + n
        elif w == 5:
            v, i = b[i:i + 4], i + 4
        else:
            raise ValueError("wire type %d" % w)
        yield f, w, v


def parse_shape(b):
    dims = []
    for f, w, v in fields(b):
        if f == 2:
            for f2, w2, v2 in fields(v):
                if f2 == 1:
                    dims.append(v2 if v2 < (1 << 62) else v2 - (1 << 64))
    return dims


def parse_tensor(b):
    out = {"dtype": 0, "shape": [], "floats": None, "ints": None}
    for f, w, v in fields(b):
        if f == 1:
            out["dtype"] = v
        elif f == 2:
            out["shape"] = parse_shape(v)
        elif f == 4:  # tensor_content
            if out["dtype"] == 1:
                out["floats"] = list(struct.unpack("<%df" % (len(v) // 4), v))
            elif out["dtype"] == 3:
                out["ints"] = list(struct.unpack("<%di" % (len(v) // 4), v))
        elif f == 5:
            out["floats"] = (out["floats"] or []) + (list(struct.unpack("<%df" % (len(v) // 4), v)) if w == 2 else [struct.unpack("<f", v)[0]])
        elif f == 7:
            if w == 2:
                j, vals = 0, []
                while j < len(v):
                    x, j = varint(v, j)
                    vals.append(x)
                out["ints"] = (out["ints"] or []) + vals
            else:
                out["ints"] = (out["ints"] or []) + [v]
    return out


def parse_attr(b):
    for f, w, v in fields(b):
        if f == 1:  # list
            ints, strs = [], []
            for f2, w2, v2 in fields(v):
                if f2 == 3:
                    if w2 == 2:
                        j = 0
                        while j < len(v2):
                            x, j = varint(v2, j)
                            ints.append(x)
                    else:
                        ints.append(v2)
                elif f2 == 2:
                    strs.append(v2.decode("latin1"))
            return {"list_i": ints} if ints else {"list_s": strs}
        if f == 2:
            return {"s": v.decode("latin1")}
        if f == 3:
            return {"i": v}
        if f == 4:
            return {"f": struct.unpack("<f", v)[0]}
        if f == 5:
            return {"b": bool(v)}
        if f == 6:
            return {"type": v}
        if f == 7:
            return {"shape": parse_shape(v)}
        if f == 8:
            return {"tensor": parse_tensor(v)}
    return {}


def main():
    raw = open(META, "rb").read()
    graph = None
    versions = {}
    for f, w, v in fields(raw):
        if f == 2:
            graph = v
        if f == 1:
            for f2, w2, v2 in fields(v):
                if f2 == 5:
                    versions["tensorflow_version"] = v2.decode()
                if f2 == 6:
                    versions["tensorflow_git_version"] = v2.decode()
    nodes = {}
    order = []
    for f, w, v in fields(graph):
        if f != 1:
            continue
        node = {"op": "", "inputs": [], "attr": {}}
        name = ""
        for f2, w2, v2 in fields(v):
            if f2 == 1:
                name = v2.decode()
            elif f2 == 2:
                node["op"] = v2.decode()
            elif f2 == 3:
                node["inputs"].append(v2.decode())
            elif f2 == 5:
                k, val = None, None
                for f3, w3, v3 in fields(v2):
                    if f3 == 1:
                        k = v3.decode()
                    elif f3 == 2:
                        val = parse_attr(v3)
                node["attr"][k] = val
        nodes[name] = node
        order.append(name)
    keep_ops = {"Conv2D", "AvgPool", "ResizeNearestNeighbor", "Maximum", "ConcatV2", "MatMul", "Sigmoid", "Mul", "Sub",
                "Reshape", "Add", "Placeholder", "VariableV2"}
    out = {"source": "HM-16.5_Test_AI/bin/model_2000000_qp30~35.dat.meta", "versions": versions, "nodes": {}}
    for name in order:
        n = nodes[name]
        if "/" in name and name.split("/")[0] in ("gradients", "Momentum", "save", "report_uninitialized_variables"):
            continue
        if name.startswith(("gradients", "Momentum", "save", "init")):
            continue
        rec = None
        if n["op"] == "Const":
            t = (n["attr"].get("value") or {}).get("tensor")
            if t:
                vals = t["floats"] if t["floats"] is not None else t["ints"]
                numel = 1
                for d in t["shape"]:
                    numel *= d
                if vals is not None and (numel <= 8 or len(set(vals)) == 1):
                    if len(vals) > 8 or (len(vals) == 1 and numel > 1):
                        vals = vals[:1]
                    rec = {"op": "Const", "dtype": t["dtype"], "shape": t["shape"], "values": vals}
                    if t["dtype"] == 1:
                        rec["bits"] = ["0x%08x" % struct.unpack("<I", struct.pack("<f", x))[0] for x in vals]
        elif n["op"] in keep_ops:
            rec = {"op": n["op"], "inputs": n["inputs"]}
            for k in ("strides", "ksize", "padding", "data_format", "transpose_a", "transpose_b", "shape", "align_corners"):
                if k in n["attr"] and n["attr"][k]:
                    a = n["attr"][k]
                    rec[k] = list(a.values())[0]
        if rec:
            out["nodes"][name] = rec
    path = os.path.join(HERE, "meta_constants.json")
    json.dump(out, open(path, "w"), indent=0, sort_keys=True)
    print("wrote", path, os.path.getsize(path), "bytes;", len(out["nodes"]), "nodes;", versions)
    for k in ("scalar", "scalar_1", "LeakyRelu/alpha", "Const", "Conv2D", "Conv2D_1", "AvgPool", "ResizeNearestNeighbor/size", "concat", "concat_1"):
        print(k, out["nodes"].get(k))


if __name__ == "__main__":
    main()
