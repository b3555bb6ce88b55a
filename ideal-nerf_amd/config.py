"""The reference's flag surface and config-file format (NeRFs/HeadNeRF/helper.py:16-138).

The reference parses ~60 flags with configargparse at import time; its config files are
``key = value`` lines whose keys are matched like abbreviated command-line options (e.g.
``N_sample=64`` selects ``--N_samples``), and unknown keys abort the run (several shipped
configs carry stale keys such as ``use_highlight``).  This module reproduces that behaviour
without the import-time global: ``load_config(path, argv)`` returns a namespace, and
``to_render_config`` hands the per-ray subset to the renderer.
"""
import argparse
import os
from types import SimpleNamespace

from .helper import RenderConfig

# (dest, type, default, action) -- the parser's surface, in declaration order
FLAGS = [
    ("config", str, None, "store"), ("expname", str, None, "store"), ("basedir", str, None, "store"),
    ("datadir", str, "./dataset/Obama", "store"), ("vis_path", str, "./dataset/Obama/run", "store"),
    ("save_path", str, "output/render/Obama-Noah/", "store"), ("evalExpr_path", str, None, "store"),
    ("mouth_rays", int, 0, "store"), ("torso_rays", int, 0, "store"), ("dim_expr", int, 0, "store"),
    ("dim_aud", int, 0, "store"), ("lc_weight", float, 0.0005, "store"), ("gt_dirs", str, "head_imgs", "store"),
    ("gpu_num", int, 0, "store"), ("num_work", int, 3, "store"), ("batch_size", int, 4, "store"),
    ("netdepth", int, 8, "store"), ("netwidth", int, 256, "store"), ("netdepth_fine", int, 8, "store"),
    ("netwidth_fine", int, 256, "store"), ("N_rand", int, 2048, "store"), ("lrate", float, 0.0008, "store"),
    ("lrate_decay", int, 500, "store"), ("chunk", int, 8192, "store"), ("netchunk", int, 65536, "store"),
    ("use_batching", bool, True, "store_false"), ("no_reload", bool, False, "store_true"),
    ("ft_path", str, None, "store"), ("N_iters", int, 90, "store"), ("N_samples", int, 64, "store"),
    ("N_importance", int, 128, "store"), ("perturb", float, 1.0, "store"),
    ("use_viewdirs", bool, True, "store_false"), ("i_embed", int, 0, "store"), ("multires", int, 10, "store"),
    ("multires_views", int, 4, "store"), ("raw_noise_std", float, 0.0, "store"),
    ("render_only", bool, False, "store_true"), ("render_test", bool, False, "store_true"),
    ("render_factor", int, 0, "store"), ("precrop_iters", int, 0, "store"), ("precrop_frac", float, 0.5, "store"),
    ("testskip", int, 8, "store"), ("white_bkgd", bool, True, "store_false"), ("half_res", bool, False, "store_true"),
    ("with_test", int, 0, "store"), ("sample_rate", float, 0.95, "store"), ("near", float, 0.3, "store"),
    ("far", float, 0.9, "store"), ("test_file", str, None, "store"), ("aud_file", str, "aud.npy", "store"),
    ("win_size", int, 16, "store"), ("smo_size", int, 8, "store"), ("nosmo_iters", int, 300000, "store"),
    ("no_ndc", bool, False, "store_true"), ("lindisp", bool, False, "store_true"), ("i_print", int, 10, "store"),
    ("i_img", int, 500, "store"), ("i_weights", int, 5000, "store"), ("i_testset", int, 1000, "store"),
    ("i_video", int, 5000, "store"),
]


def make_parser() -> argparse.ArgumentParser:
    p = argparse.ArgumentParser(allow_abbrev=True)
    for dest, typ, default, action in FLAGS:
        if action == "store":
            p.add_argument("--" + dest, type=typ, default=default)
        else:
            p.add_argument("--" + dest, action=action)
    return p


def config_lines_to_argv(text: str):
    """``key = value`` lines -> the argv configargparse would synthesise (``#``/``;`` comments
    and blank lines skipped; a flag without value, or ``true``, switches a store_true/false)."""
    argv = []
    for raw in text.splitlines():
        line = raw.strip()
        if not line or line[0] in "#;":
            continue
        if "=" in line:
            key, val = line.split("=", 1)
        elif " " in line:
            key, val = line.split(None, 1)
        else:
            key, val = line, ""
        key, val = key.strip(), val.strip()
        if val.lower() in ("", "true"):
            argv += ["--" + key] if val.lower() == "true" or val == "" else []
        elif val.lower() == "false":
            continue
        else:
            argv += ["--" + key, val]
    return argv


def load_config(path: str = None, argv=None, text: str = None) -> SimpleNamespace:
    """Defaults <- config file <- command line, as the reference resolves them.  Unknown or
    ambiguous keys raise ValueError (the reference's parser exits)."""
    parser = make_parser()
    file_argv = []
    if text is None and path is not None:
        with open(path) as f:
            text = f.read()
    if text is not None:
        file_argv = config_lines_to_argv(text)
    full = file_argv + list(argv or [])
    try:
        ns, unknown = parser.parse_known_args(full)
    except SystemExit as e:  # ambiguous abbreviation / bad type
        raise ValueError(f"config rejected by the flag parser (exit {e.code})") from None
    if unknown:
        raise ValueError(f"unrecognized config keys: {[u for u in unknown if u.startswith('--')]}")
    out = SimpleNamespace(**vars(ns))
    out.config = path
    return out


def to_render_config(ns, dim_latent: int = 32) -> RenderConfig:
    """The subset of flags the per-ray path reads (audio_exp_nerf.py:213-226,297-364)."""
    return RenderConfig(netdepth=ns.netdepth, netwidth=ns.netwidth, dim_aud=ns.dim_aud, dim_expr=ns.dim_expr,
                        dim_latent=dim_latent, win_size=ns.win_size, smo_size=ns.smo_size, nosmo_iters=ns.nosmo_iters,
                        N_samples=ns.N_samples, N_importance=ns.N_importance, perturb=ns.perturb, chunk=ns.chunk,
                        netchunk=ns.netchunk, multires=ns.multires, multires_views=ns.multires_views,
                        use_viewdirs=ns.use_viewdirs, near=ns.near, far=ns.far, lc_weight=ns.lc_weight)


def write_config(ns, config_text: str = None):
    """helper.py:371-384: <basedir>/<expname>/args.txt (sorted ``k = v``) and a copy of the config."""
    d = os.path.join(ns.basedir, ns.expname)
    os.makedirs(d, exist_ok=True)
    with open(os.path.join(d, "args.txt"), "w") as f:
        for k in sorted(vars(ns)):
            f.write(f"{k} = {getattr(ns, k)}\n")
    if ns.config is not None:
        if config_text is None:
            config_text = open(ns.config).read()
        with open(os.path.join(d, "config.txt"), "w") as f:
            f.write(config_text)
