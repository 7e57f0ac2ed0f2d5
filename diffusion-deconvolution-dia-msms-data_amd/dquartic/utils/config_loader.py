"""JSON training configuration -- same schema, defaults and override rules as the reference's
``dquartic/utils/config_loader.py`` (:4-57 load with CLI overrides, :60-119 default-config writer).
Additions (all optional, defaulting to the reference behaviour): ``data.synthetic`` = {"n_windows", "RT", "MZ"} selects the
built-in synthetic dataset instead of files; integer-like CLI overrides that arrive as strings (``--batch-size``,
``--threads`` have no click type in the reference, cli.py:39,42) are coerced to int."""
import copy
import json

DEFAULT_CONFIG = {
    "data": {"parquet_directory": "data/", "ms2_data_path": None, "ms1_data_path": None, "normalize": "minmax"},
    "model": {
        "checkpoint_path": "best_model.ckpt", "num_epochs": 10000, "warmup_epochs": 5, "batch_size": 1,
        "learning_rate": 0.00001, "num_timesteps": 1000, "beta_schedule_type": "cosine", "pred_type": "eps",
        "auto_normalize": True, "ms1_loss_weight": 0.0, "use_model": "UNet1d",
        "CustomTransformer": {"input_dim": 40000, "hidden_dim": 1024, "num_heads": 8, "num_layers": 8},
        "UNet1d": {"dim": 4, "channels": 1, "dim_mults": [1, 2, 2, 3, 3, 4, 4], "conditional": True, "init_cond_channels": 1,
                   "attn_cond_channels": 1, "tfer_dim_mult": 620, "downsample_dim": 40000, "simple": True},
    },
    "wandb": {"use_wandb": True, "wandb_project": "dquartic", "wandb_name": None, "wandb_id": None, "wandb_resume": None,
              "wandb_architecture": "DDIM(UNet1d)", "wandb_dataset": "MS2", "wandb_mode": "offline"},
    "threads": 4,
}

_OVERRIDES = {  # kwarg -> (section, key, coerce)
    "parquet_directory": ("data", "parquet_directory", None),
    "ms2_data_path": ("data", "ms2_data_path", None),
    "ms1_data_path": ("data", "ms1_data_path", None),
    "batch_size": ("model", "batch_size", int),
    "checkpoint_path": ("model", "checkpoint_path", None),
    "use_wandb": ("wandb", "use_wandb", None),
    "threads": (None, "threads", int),
}


def load_train_config(config_path: str, **kwargs):
    with open(config_path, "r") as f:
        cfg = json.load(f)
    for key in ("parquet_directory", "ms2_data_path", "ms1_data_path"):
        cfg["data"].setdefault(key, None)
    for name, (section, key, coerce) in _OVERRIDES.items():
        val = kwargs.get(name)
        if val is None:
            continue
        if coerce is not None:
            val = coerce(val)
        if section is None:
            cfg[key] = val
        else:
            cfg[section][key] = val
    cfg["model"]["batch_size"] = int(cfg["model"]["batch_size"])
    cfg["threads"] = int(cfg.get("threads", 0))
    return cfg


def generate_train_config(config_path: str):
    with open(config_path, "w") as f:
        json.dump(copy.deepcopy(DEFAULT_CONFIG), f, indent=4)
