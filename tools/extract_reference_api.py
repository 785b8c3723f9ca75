"""Build-container only: pin the HOST LOGIC of the hot path to the reference TEXT.

`python tools/extract_reference_api.py` parses /root/reference/chambers/** with `ast` (no `import chambers`, no
TensorFlow - the package imports TF at module top and TF is not installed here) and writes
tests/golden/reference_api.json, a data fixture:

  * every class of the SURVEY 8a / 8f files: bases, Keras registration package, `__init__` argument names and literal
    defaults, the key set of the `config` dict literal inside `get_config`, whether `get_config` merges the base
    config, the names given to `add_weight` in `build`;
  * every module-level builder function: argument names and literal defaults (VisionTransformer, ViT*/DeiT*);
  * the zoo constants of each ViT*/DeiT* builder (its local assignments and the literal keywords of the call it makes);
  * `_AUTO_AUGMENT_POLICY_V0`, the `_INTERPOLATION_MODE / _FILL_MODE / _FILL_VALUE / _MAX_MAGNITUDE` constants, the
    transform -> magnitude-function map of `_get_transform`, the op order of `RandAugment.__init__`;
  * OUTPUTS of reference code that is pure Python, obtained by compiling the extracted function definitions alone
    (their source segments, nothing else of the module) in an empty namespace: `_magnitude_to_*_kwargs` for magnitudes
    0..10, `_are_weights_pretrained` / `_get_model_info` over the `WEIGHTS_HASHES` table, and
    `WeightDecayExtension._is_decay_allowed` over a list of Keras variable names x pattern sets.

The fixture holds data (names, literals, input -> output pairs), no reference source text.  tests/test_reference_api.py
compares BOTH `chambers_amd` and `oracle/` with it.  /root/reference does not exist on the GPU box: only this script
reads it, and only here.
"""
import ast
import json
import os
import re
import sys

REF = os.environ.get("CHB_REFERENCE_ROOT", "/root/reference/chambers")
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "reference_api.json")

FILES = [
    "augmentations/augmentation_schemes.py",
    "augmentations/image_augmentations.py",
    "layers/attention.py",
    "layers/transformer.py",
    "layers/embedding.py",
    "layers/normalization.py",
    "models/backbones/vision_transformer.py",
    "optimizers.py",
    "schedules.py",
    "activations.py",
    "miners.py",
    "losses/metric_learning.py",
]

# names the decay filters are evaluated on (Keras `var.name` strings of a ViT, as the reference's layers would name them)
DECAY_NAMES = [
    "patch_embeddings/embedding/kernel:0", "patch_embeddings/embedding/bias:0", "add_cls_token/embeddings:0",
    "add_dist_token/embeddings:0", "pos_embedding/embeddings:0",
    "encoder/encoder_layer/multi_head_attention/w_query:0", "encoder/encoder_layer/multi_head_attention/b_query:0",
    "encoder/encoder_layer/multi_head_attention/w_key:0", "encoder/encoder_layer/multi_head_attention/b_key:0",
    "encoder/encoder_layer/multi_head_attention/w_value:0", "encoder/encoder_layer/multi_head_attention/b_value:0",
    "encoder/encoder_layer/multi_head_attention/w_projection:0", "encoder/encoder_layer/multi_head_attention/b_projection:0",
    "encoder/encoder_layer/layer_normalization/gamma:0", "encoder/encoder_layer/layer_normalization/beta:0",
    "encoder/encoder_layer/layer_normalization_1/gamma:0", "encoder/encoder_layer/dense/kernel:0", "encoder/encoder_layer/dense/bias:0",
    "encoder/encoder_layer_1/dense_2/kernel:0", "encoder/encoder_layer_1/dense_3/bias:0",
    "encoder/layer_normalization_4/gamma:0", "encoder/layer_normalization_4/beta:0",
    "feature/kernel:0", "feature/bias:0", "predictions/kernel:0", "predictions/bias:0", "predictions_dist/kernel:0",
]
DECAY_FILTERS = [
    {"decay_include": None, "decay_exclude": None},
    {"decay_include": ["kernel"], "decay_exclude": None},
    {"decay_include": ["kernel:0$", "/w_"], "decay_exclude": None},
    {"decay_include": None, "decay_exclude": ["bias", "/b_", "layer_normalization", "embeddings"]},
    {"decay_include": None, "decay_exclude": ["gamma", "beta"]},
    {"decay_include": [r"encoder_layer_1/"], "decay_exclude": None},
    {"decay_include": [], "decay_exclude": None},
    {"decay_include": None, "decay_exclude": []},
]


def literal(node):
    """A JSON-able literal for an ast node, or {"expr": source} for anything that is not a literal."""
    try:
        v = ast.literal_eval(node)
    except Exception:
        return {"expr": ast.unparse(node)}
    return to_json(v)


def to_json(v):
    if isinstance(v, tuple):
        return [to_json(x) for x in v]
    if isinstance(v, list):
        return [to_json(x) for x in v]
    if isinstance(v, dict):
        return {str(k): to_json(x) for k, x in v.items()}
    return v


def signature(fn):
    a = fn.args
    pos = [x.arg for x in a.posonlyargs + a.args]
    defaults = {}
    for name, d in zip(pos[len(pos) - len(a.defaults):], a.defaults):
        defaults[name] = literal(d)
    kwonly = [x.arg for x in a.kwonlyargs]
    for x, d in zip(a.kwonlyargs, a.kw_defaults):
        if d is not None:
            defaults[x.arg] = literal(d)
    return {"args": [p for p in pos if p not in ("self", "cls")], "kwonly": kwonly, "defaults": defaults,
            "varargs": a.vararg.arg if a.vararg else None, "varkw": a.kwarg.arg if a.kwarg else None}


def registered_package(cls):
    for d in cls.decorator_list:
        if isinstance(d, ast.Call) and "register_keras_serializable" in ast.unparse(d.func):
            for kw in d.keywords:
                if kw.arg == "package":
                    return literal(kw.value)
            if d.args:
                return literal(d.args[0])
            return "Custom"
    return None


def config_keys(fn):
    """Keys of the dict literal assigned to `config` in get_config (+ whether the base config is merged in)."""
    keys, merges = None, False
    for node in ast.walk(fn):
        if isinstance(node, ast.Assign) and any(isinstance(t, ast.Name) and t.id == "config" for t in node.targets) \
                and isinstance(node.value, ast.Dict):
            keys = [literal(k) for k in node.value.keys]
        if isinstance(node, ast.Call) and isinstance(node.func, ast.Attribute) and node.func.attr == "get_config" \
                and isinstance(node.func.value, ast.Call) and getattr(node.func.value.func, "id", "") == "super":
            merges = True
    return keys, merges


def add_weight_names(cls):
    names = []
    for node in ast.walk(cls):
        if isinstance(node, ast.Call) and isinstance(node.func, ast.Attribute) and node.func.attr == "add_weight":
            for kw in node.keywords:
                if kw.arg == "name":
                    names.append(literal(kw.value))
    return names


def describe_class(cls):
    out = {"bases": [ast.unparse(b) for b in cls.bases], "registered_package": registered_package(cls), "line": cls.lineno,
           "methods": [n.name for n in cls.body if isinstance(n, ast.FunctionDef)]}
    for n in cls.body:
        if isinstance(n, ast.FunctionDef) and n.name == "__init__":
            out["init"] = signature(n)
        if isinstance(n, ast.FunctionDef) and n.name == "get_config":
            keys, merges = config_keys(n)
            out["get_config_keys"] = keys
            out["get_config_merges_base"] = merges
    w = add_weight_names(cls)
    if w:
        out["add_weight_names"] = w
    return out


def find(tree, kind, name):
    for n in ast.walk(tree):
        if isinstance(n, kind) and getattr(n, "name", None) == name:
            return n
    raise KeyError(name)


def run_extracted(src, nodes, extra_globals=None):
    """Compile ONLY the given top-level nodes' source segments in a fresh namespace (pure-Python reference functions)."""
    ns = {"__builtins__": {"int": int, "float": float, "str": str, "len": len, "range": range, "tuple": tuple, "list": list,
                           "dict": dict, "isinstance": isinstance, "None": None, "True": True, "False": False}}
    ns.update(extra_globals or {})
    for n in nodes:
        exec(compile(ast.get_source_segment(src, n), "<reference segment>", "exec"), ns)
    return ns


def main():
    if not os.path.isdir(REF):
        sys.exit("reference tree not found at %s (this script runs in the build container only)" % REF)
    fixture = {"_generated_by": "tools/extract_reference_api.py (ast over /root/reference/chambers, no TensorFlow import)",
               "classes": {}, "functions": {}}
    trees, sources = {}, {}
    for rel in FILES:
        with open(os.path.join(REF, rel)) as f:
            sources[rel] = f.read()
        trees[rel] = ast.parse(sources[rel])
        fixture["classes"][rel] = {n.name: describe_class(n) for n in trees[rel].body if isinstance(n, ast.ClassDef)}
        fixture["functions"][rel] = {n.name: dict(signature(n), line=n.lineno) for n in trees[rel].body if isinstance(n, ast.FunctionDef)}
        # classes defined inside functions (extend_with_weight_decay) are not part of the hot path

    # ---- augmentation schemes -------------------------------------------------------------------------------------
    rel = "augmentations/augmentation_schemes.py"
    tree, src = trees[rel], sources[rel]
    consts = {}
    const_nodes = []
    for n in tree.body:
        if isinstance(n, ast.Assign) and len(n.targets) == 1 and isinstance(n.targets[0], ast.Name) and n.targets[0].id.startswith("_"):
            consts[n.targets[0].id] = literal(n.value)
            const_nodes.append(n)
    fixture["augmentation_constants"] = {k: v for k, v in consts.items() if k != "_AUTO_AUGMENT_POLICY_V0"}
    fixture["auto_augment_policy_v0"] = consts["_AUTO_AUGMENT_POLICY_V0"]
    get_transform = find(tree, ast.FunctionDef, "_get_transform")
    fn_map = {}
    for node in ast.walk(get_transform):
        if isinstance(node, ast.Assign) and getattr(node.targets[0], "id", "") == "magnitude_fn_map":
            for k, v in zip(node.value.keys, node.value.values):
                fn_map[literal(k)] = v.id if isinstance(v, ast.Name) else "lambda:" + ast.unparse(v.body)
    fixture["magnitude_fn_map"] = fn_map
    mag_fns = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name.startswith("_magnitude_to_")]
    ns = run_extracted(src, const_nodes + mag_fns)
    kwargs = {}
    for name, fn in fn_map.items():
        kwargs[name] = [to_json(ns[fn](m)) if not fn.startswith("lambda:") else {} for m in range(11)]
    fixture["magnitude_kwargs"] = kwargs          # [transform][magnitude 0..10] -> constructor kwargs
    ra = find(tree, ast.ClassDef, "RandAugment")
    order = []
    for node in ast.walk(find(ra, ast.FunctionDef, "__init__")):
        if isinstance(node, ast.Call) and getattr(node.func, "id", "") == "_get_transform":
            order.append(literal(node.args[0]))
    fixture["randaugment_ops"] = order
    aa = find(tree, ast.ClassDef, "AutoAugment")
    fixture["autoaugment_choice"] = {}
    for node in ast.walk(find(aa, ast.FunctionDef, "__init__")):
        if isinstance(node, ast.Call) and ast.unparse(node.func).endswith("RandomChoice"):
            fixture["autoaugment_choice"] = {kw.arg: literal(kw.value) for kw in node.keywords}

    # ---- ViT zoo --------------------------------------------------------------------------------------------------
    rel = "models/backbones/vision_transformer.py"
    tree, src = trees[rel], sources[rel]
    zoo = {}
    for n in tree.body:
        if isinstance(n, ast.FunctionDef) and re.match(r"(ViT|DeiT)[A-Z]\d+$", n.name):
            local = {}
            call_kw, callee = {}, None
            for node in ast.walk(n):
                if isinstance(node, ast.Assign) and isinstance(node.targets[0], ast.Name):
                    v = literal(node.value)
                    if not isinstance(v, dict):
                        local[node.targets[0].id] = v
                if isinstance(node, ast.Call) and getattr(node.func, "id", "") in ("VisionTransformer", "DistilledVisionTransformer"):
                    callee = node.func.id
                    for kw in node.keywords:
                        v = literal(kw.value)
                        call_kw[kw.arg] = local.get(kw.value.id, v) if isinstance(kw.value, ast.Name) else v
            zoo[n.name] = {"builder": callee, "constants": local,
                           "call_literals": {k: v for k, v in call_kw.items() if not isinstance(v, dict)},
                           "defaults": signature(n)["defaults"]}
    fixture["zoo"] = zoo
    hashes = find_assign(tree, "WEIGHTS_HASHES")
    table = ast.literal_eval(hashes.value)
    fixture["pretrained_weight_names"] = {m: sorted(w) for m, w in table.items()}
    ns = run_extracted(src, [hashes, find(tree, ast.FunctionDef, "_are_weights_pretrained"), find(tree, ast.FunctionDef, "_get_model_info")])
    info = {}
    for m in list(table) + ["custom"]:
        for wname in sorted({w for ws in table.values() for w in ws}) + [None, "some/path.h5"]:
            d, f = ns["_get_model_info"](wname, m)
            info.setdefault(m, {})[str(wname)] = {"pretrained": bool(ns["_are_weights_pretrained"](wname, m)), "default_size": d, "has_feature": bool(f)}
    fixture["model_info"] = info
    fixture["preprocess_input"] = {}
    for n in tree.body:
        if isinstance(n, ast.Assign) and getattr(n.targets[0], "id", "") == "preprocess_input" and isinstance(n.value, ast.Call):
            fixture["preprocess_input"] = {"class": ast.unparse(n.value.func), "kwargs": {kw.arg: literal(kw.value) for kw in n.value.keywords},
                                           "args": [literal(a) for a in n.value.args]}
    # layer names given inside the builders (name= keywords with literal strings), in source order
    for fn_name in ("VisionTransformer", "DistilledVisionTransformer", "_pool"):
        fn = find(tree, ast.FunctionDef, fn_name)
        names = []
        for node in sorted((x for x in ast.walk(fn) if isinstance(x, ast.Call)), key=lambda x: (x.lineno, x.col_offset)):
            for kw in node.keywords:
                if kw.arg == "name":
                    names.append([ast.unparse(node.func), literal(kw.value) if not isinstance(kw.value, ast.BinOp) else {"expr": ast.unparse(kw.value)}])
        fixture.setdefault("builder_layer_names", {})[fn_name] = names
    # numeric literals of the layers the builders construct (epsilon, strides, ...) as keyword -> literal per call
    calls = {}
    for fn_name in ("VisionTransformer", "DistilledVisionTransformer"):
        fn = find(tree, ast.FunctionDef, fn_name)
        rows = []
        for node in sorted((x for x in ast.walk(fn) if isinstance(x, ast.Call)), key=lambda x: (x.lineno, x.col_offset)):
            f = ast.unparse(node.func)
            if f.split(".")[-1] in ("Conv2D", "Encoder", "Dense", "Dropout", "ConcatEmbedding", "LearnedEmbedding1D", "Reshape", "Activation"):
                rows.append([f.split(".")[-1], {kw.arg: literal(kw.value) for kw in node.keywords if kw.arg}])
        calls[fn_name] = rows
    fixture["builder_calls"] = calls

    # ---- encoder layer internals: the sub-layers EncoderLayer / Encoder construct, with their literal keywords -----------
    rel = "layers/transformer.py"
    tree = trees[rel]
    sub = {}
    for cname in ("EncoderLayer", "Encoder"):
        init = find(find(tree, ast.ClassDef, cname), ast.FunctionDef, "__init__")
        rows = []
        for node in sorted((x for x in ast.walk(init) if isinstance(x, ast.Call)), key=lambda x: (x.lineno, x.col_offset)):
            f = ast.unparse(node.func).split(".")[-1]
            if f in ("MultiHeadAttention", "Dense", "Dropout", "LayerNormalization", "EncoderLayer", "Add"):
                rows.append([f, {kw.arg: literal(kw.value) for kw in node.keywords if kw.arg}])
        sub[cname] = rows
    fixture["encoder_sublayers"] = sub

    # ---- AdamW decay filter (pure Python: `re` on var.name) ---------------------------------------------------------
    rel = "optimizers.py"
    tree, src = trees[rel], sources[rel]
    wde = find(tree, ast.ClassDef, "WeightDecayExtension")
    allowed = find(wde, ast.FunctionDef, "_is_decay_allowed")
    seg = ast.get_source_segment(src, allowed)
    import textwrap
    ns = {"re": re}
    exec(compile(textwrap.dedent(seg), "<reference segment>", "exec"), ns)

    class _Self:
        pass

    class _Var:
        def __init__(self, name):
            self.name = name
    rows = []
    for flt in DECAY_FILTERS:
        s = _Self()
        s.decay_include, s.decay_exclude = flt["decay_include"], flt["decay_exclude"]
        rows.append({"filter": flt, "allowed": {n: bool(ns["_is_decay_allowed"](s, _Var(n))) for n in DECAY_NAMES}})
    fixture["adamw_is_decay_allowed"] = rows
    # the exclusive-arguments check of the constructor: which exception type is raised for include + exclude
    init = find(wde, ast.FunctionDef, "__init__")
    raises = [ast.unparse(n.exc.func) for n in ast.walk(init) if isinstance(n, ast.Raise) and isinstance(n.exc, ast.Call)]
    fixture["adamw_init_raises"] = raises

    # ---- activations: the numeric literals of gelu ---------------------------------------------------------------------
    rel = "activations.py"
    gelu = find(trees[rel], ast.FunctionDef, "gelu")
    fixture["gelu"] = {"signature": signature(gelu),
                       "float_literals": sorted({n.value for n in ast.walk(gelu) if isinstance(n, ast.Constant) and isinstance(n.value, float)})}

    with open(OUT, "w") as f:
        json.dump(fixture, f, indent=1, sort_keys=True)
        f.write("\n")
    print("wrote", OUT, os.path.getsize(OUT), "bytes")


def find_assign(tree, name):
    for n in tree.body:
        if isinstance(n, ast.Assign) and getattr(n.targets[0], "id", "") == name:
            return n
    raise KeyError(name)


if __name__ == "__main__":
    main()
