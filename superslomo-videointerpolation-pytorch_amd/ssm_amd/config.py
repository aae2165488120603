"""ini handling: the reference threads one RawConfigParser (`cfg`) through every
constructor; this loads the same files and applies documented overrides."""
import configparser
import os

CONFIG_DIR = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "configs")


def load_config(name_or_path="superslomo_original.ini", overrides=None):
    """overrides: {(section, key): value}.  The benchmark/test harness only ever overrides
    paths, LOADPREV and FREEZE (random-init weights: none are distributed)."""
    path = name_or_path if os.path.exists(name_or_path) else os.path.join(CONFIG_DIR, name_or_path)
    cfg = configparser.RawConfigParser()
    if not cfg.read(path):
        raise FileNotFoundError(path)
    for (sec, key), val in (overrides or {}).items():
        cfg.set(sec, key, str(val))
    return cfg


def synthetic_weight_overrides():
    return {("STAGE1", "LOADPREV"): "FALSE", ("STAGE2", "LOADPREV"): "FALSE"}
