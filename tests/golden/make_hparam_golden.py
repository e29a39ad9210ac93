"""Generates tests/golden/hparam_golden.json by IMPORTING the reference's own
athena/utils/hparam.py (it has no TensorFlow dependency) and recording what it does for a list of
cases; also copies the shipped JSON configs of the ASR recipes (data files) next to it.
Run in the build container only: /root/reference does not exist on the GPU box.

    python tests/golden/make_hparam_golden.py
"""
import importlib.util
import json
import os
import shutil

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))

spec = importlib.util.spec_from_file_location("ref_hparam", os.path.join(REF, "athena/utils/hparam.py"))
ref = importlib.util.module_from_spec(spec)
spec.loader.exec_module(ref)

# default_config dicts of the classes on the training path (values as in the reference)
DEFAULTS = {
    "solver": {"clip_norm": 100.0, "log_interval": 10, "enable_tf_function": True},
    "warmup_adam": {"d_model": 512, "warmup_steps": 8000, "k": 0.5, "decay_steps": 100000, "decay_rate": 1.0},
    "speech_transformer": {"return_encoder_output": False, "num_filters": 512, "d_model": 512, "num_heads": 8,
                           "num_encoder_layers": 12, "num_decoder_layers": 6, "dff": 1280, "rate": 0.1,
                           "schedual_sampling_rate": 0.9, "label_smoothing_rate": 0.0},
    "mtl": {"model": "speech_transformer", "model_config": {"return_encoder_output": True}, "mtl_weight": 0.5},
    "mixed": {"a": 1, "b": 2.5, "c": "x", "d": [1, 2, 3], "e": None, "f": False, "g": [0.5, 1.5]},
}

OVERRIDES = [
    ("solver", {"clip_norm": 5, "log_interval": 100}),
    ("solver", {"enable_tf_function": False}),
    ("solver", {"clip_norm": "100"}),
    ("solver", {"unknown_key": 1}),
    ("solver", {"log_interval": 2.5}),
    ("solver", None),
    ("warmup_adam", {"d_model": 256, "warmup_steps": 25000, "k": 1.0, "decay_steps": 100000000, "decay_rate": 0.1}),
    ("warmup_adam", {"k": 1}),
    ("warmup_adam", {"d_model": 256.0}),
    ("speech_transformer", {"return_encoder_output": True, "num_filters": 256, "d_model": 256, "num_heads": 4,
                            "num_encoder_layers": 12, "num_decoder_layers": 6, "dff": 1024, "rate": 0.1,
                            "label_smoothing_rate": 0.1}),
    ("speech_transformer", {"rate": 1}),
    ("speech_transformer", {"num_heads": "4"}),
    ("speech_transformer", {"num_head": 4}),
    ("mtl", {"model_config": {"d_model": 256, "extra": [1, 2]}, "mtl_weight": 0.3}),
    ("mtl", {"model": 3}),
    ("mixed", {"a": 2, "b": 3, "c": "yy", "d": [4, 5], "e": "now set", "f": True, "g": [1, 2]}),
    ("mixed", {"a": True}),
    ("mixed", {"f": 1}),
    ("mixed", {"d": 7}),
    ("mixed", {"d": [1.5]}),
    ("mixed", {"c": 3}),
]

PARSE = [
    ("mixed", "a=3,b=0.25,c=hello"),
    ("mixed", "d=[7,8,9],f=true"),
    ("mixed", "d[1]=5"),
    ("mixed", "g=[1,2.5]"),
    ("mixed", "zzz=1"),
    ("mixed", "a=notint"),
]


def snapshot(hp):
    return json.loads(hp.to_json())


def main():
    out = {"defaults": DEFAULTS, "override": [], "parse": []}
    for name, cfg in OVERRIDES:
        rec = {"defaults": name, "config": cfg}
        try:
            hp = ref.register_and_parse_hparams(dict(DEFAULTS[name]), cfg)
            rec["values"] = snapshot(hp)
        except Exception as e:                      # the reference's error class is part of the contract
            rec["raises"] = type(e).__name__
        out["override"].append(rec)
    for name, text in PARSE:
        rec = {"defaults": name, "text": text}
        try:
            hp = ref.register_and_parse_hparams(dict(DEFAULTS[name]), None)
            hp.parse(text)
            rec["values"] = snapshot(hp)
        except Exception as e:
            rec["raises"] = type(e).__name__
        out["parse"].append(rec)
    with open(os.path.join(HERE, "hparam_golden.json"), "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
    for recipe in ("timit", "aishell", "librispeech", "hkust"):
        src = os.path.join(REF, "examples/asr", recipe, "configs/mtl_transformer_sp.json")
        if os.path.exists(src):
            shutil.copy(src, os.path.join(HERE, "configs", "%s_mtl_transformer_sp.json" % recipe))
    print("wrote", len(out["override"]), "override and", len(out["parse"]), "parse cases")


if __name__ == "__main__":
    main()
