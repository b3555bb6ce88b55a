"""The reference's flag surface and config-file format (NeRFs/HeadNeRF/helper.py:16-138).

The reference parses ~60 flags with configargparse at import time; its config files are
``key = value`` lines whose keys are matched like abbreviated command-line options (e.g.
``N_sample=64`` selects ``--N_samples``), and unknown keys abort the run (several shipped
configs carry stale keys such as ``use_highlight``).  This module reproduces that behaviour:
``load_config(path, argv)`` returns a namespace, ``to_render_config`` hands the per-ray subset to the
renderer, and ``ConfigParser`` is what ``helper.config_parser()`` returns -- ``parse_args()`` resolves
defaults <- ``--config`` file <- command line like configargparse and records the result as the
PROCESS'S flags (``current_args``), which a ``Network`` constructed without ``args=`` reads exactly as
the reference's classes read their import-time ``args`` global (helper.py:141-142,
audio_exp_nerf.py:25-26,199-226).
"""
import argparse
import logging
import os
import sys
from types import SimpleNamespace

from .helper import RenderConfig

logger = logging.getLogger("adnerf")

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


# NeRFs/TorsoNeRF/run_nerf_helpers.py:231-365 -- the torso stack has its own parser: other defaults (chunk 1024, dim_aud 64,
# testskip 1, lrate 5e-4), `dim_aud_body`, no `dim_expr` (the head pair's 79 is a literal, train_torso.py:203) and no `lindisp`
TORSO_FLAGS = [
    ("config", str, None, "store"), ("expname", str, None, "store"), ("basedir", str, "./logs/", "store"),
    ("datadir", str, "dataset/Obama", "store"), ("vis_path", str, "./dataset/Obama/run", "store"),
    ("save_path", str, "output/render/Obama-Noah/", "store"), ("evalExpr_path", str, None, "store"),
    ("use_highlight", bool, False, "store_true"), ("lc_weight", float, 0.0005, "store"), ("gt_dirs", str, "head_imgs", "store"),
    ("gpu_num", int, 0, "store"), ("num_work", int, 3, "store"), ("batch_size", int, 4, "store"),
    ("netdepth", int, 8, "store"), ("netwidth", int, 256, "store"), ("netdepth_fine", int, 8, "store"),
    ("netwidth_fine", int, 256, "store"), ("N_rand", int, 2048, "store"), ("lrate", float, 5e-4, "store"),
    ("lrate_decay", int, 500, "store"), ("chunk", int, 1024, "store"), ("netchunk", int, 65536, "store"),
    ("use_batching", bool, True, "store_false"), ("no_reload", bool, False, "store_true"),
    ("ft_path", str, None, "store"), ("N_iters", int, 400000, "store"), ("N_samples", int, 64, "store"),
    ("N_importance", int, 128, "store"), ("perturb", float, 1.0, "store"),
    ("use_viewdirs", bool, True, "store_false"), ("i_embed", int, 0, "store"), ("multires", int, 10, "store"),
    ("multires_views", int, 4, "store"), ("raw_noise_std", float, 0.0, "store"),
    ("render_only", bool, False, "store_true"), ("render_test", bool, False, "store_true"),
    ("render_factor", int, 0, "store"), ("precrop_iters", int, 0, "store"), ("precrop_frac", float, 0.5, "store"),
    ("dataset_type", str, "audface", "store"), ("testskip", int, 1, "store"), ("shape", str, "greek", "store"),
    ("white_bkgd", bool, True, "store_false"), ("half_res", bool, False, "store_true"),
    ("with_test", int, 0, "store"), ("dim_aud", int, 64, "store"), ("dim_aud_body", int, 64, "store"),
    ("sample_rate", float, 0.95, "store"), ("near", float, 0.3, "store"), ("far", float, 0.9, "store"),
    ("test_pose_file", str, "transforms_exp_val.json", "store"), ("aud_file", str, "aud.npy", "store"),
    ("win_size", int, 16, "store"), ("smo_size", int, 8, "store"), ("test_size", int, -1, "store"),
    ("aud_start", int, 0, "store"), ("test_save_folder", str, "test_aud_rst", "store"), ("i_print", int, 100, "store"),
    ("i_img", int, 500, "store"), ("i_weights", int, 10000, "store"), ("i_testset", int, 10000, "store"),
    ("i_video", int, 50000, "store"),
]
FLAG_TABLES = {"head": FLAGS, "torso": TORSO_FLAGS}


def make_parser(kind: str = "head") -> argparse.ArgumentParser:
    p = argparse.ArgumentParser(allow_abbrev=True)
    for dest, typ, default, action in FLAG_TABLES[kind]:
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


def load_config(path: str = None, argv=None, text: str = None, kind: str = "head") -> SimpleNamespace:
    """Defaults <- config file <- command line, as the reference resolves them.  Unknown or
    ambiguous keys raise ValueError (the reference's parser exits)."""
    parser = make_parser(kind)
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
    """The subset of flags the per-ray path reads (audio_exp_nerf.py:213-226,297-364).  A torso namespace has no
    `dim_expr` (the head pair's 79 is a literal upstream, train_torso.py:203) and no `nosmo_iters`."""
    return RenderConfig(netdepth=ns.netdepth, netwidth=ns.netwidth, dim_aud=ns.dim_aud, dim_expr=getattr(ns, "dim_expr", 79),
                        dim_latent=dim_latent, win_size=ns.win_size, smo_size=ns.smo_size,
                        nosmo_iters=getattr(ns, "nosmo_iters", 300000),
                        N_samples=ns.N_samples, N_importance=ns.N_importance, perturb=ns.perturb, chunk=ns.chunk,
                        netchunk=ns.netchunk, multires=ns.multires, multires_views=ns.multires_views,
                        use_viewdirs=ns.use_viewdirs, near=ns.near, far=ns.far, lc_weight=ns.lc_weight)


# ---- the process's flags: what the reference keeps in its import-time `args` global ------------------------------
_current = {}          # kind -> namespace of the last ConfigParser(kind).parse_args() / set_current_args()
_warned = set()


class ConfigParser:
    """What `helper.config_parser()` returns (helper.py:16-138; TorsoNeRF: run_nerf_helpers.py:231-365): the part of
    configargparse.ArgumentParser the reference's scripts use.  `parse_args()` reads `sys.argv[1:]` unless given a list,
    takes `--config <file>` as a file of `key = value` lines ranked below the command line, matches abbreviated keys
    like the upstream parser (`N_sample = 64` selects `--N_samples`), EXITS on unknown keys exactly like upstream
    (`SystemExit`), and records the namespace as the process's flags."""

    def __init__(self, kind: str = "head"):
        self.kind = kind
        self._parser = make_parser(kind)

    def add_argument(self, *a, **kw):
        kw.pop("is_config_file", None)
        return self._parser.add_argument(*a, **kw)

    def _resolve(self, argv):
        argv = list(sys.argv[1:] if argv is None else argv)
        path = None
        for i, a in enumerate(argv):
            if a == "--config" and i + 1 < len(argv):
                path = argv[i + 1]
            elif a.startswith("--config="):
                path = a.split("=", 1)[1]
        file_argv = []
        if path is not None:
            with open(path) as f:
                file_argv = config_lines_to_argv(f.read())
        return file_argv + argv

    def parse_known_args(self, args=None, namespace=None):
        ns, unknown = self._parser.parse_known_args(self._resolve(args), namespace)
        set_current_args(ns, self.kind)
        return ns, unknown

    def parse_args(self, args=None, namespace=None):
        ns = self._parser.parse_args(self._resolve(args), namespace)
        set_current_args(ns, self.kind)
        return ns


def set_current_args(ns, kind: str = "head"):
    """Make `ns` the process's flags of `kind` (what `parse_args()` does); None forgets them."""
    if ns is None:
        _current.pop(kind, None)
    else:
        _current[kind] = ns
    return ns


def current_args(kind: str = "head", parse_argv: bool = True):
    """The process's flags: the last parsed namespace; failing that -- like the reference, which parses `sys.argv` when
    its modules are imported -- the command line, if it names a `--config`; else None."""
    ns = _current.get(kind)
    if ns is None and parse_argv and any(a == "--config" or a.startswith("--config=") for a in sys.argv[1:]):
        # (implicit path: flags this parser does not know -- the caller's own, or stale keys of the file -- are ignored here;
        #  `config_parser().parse_args()` and `helper.args` stay as strict as upstream's parser and end the run on them)
        ns, _ = ConfigParser(kind).parse_known_args()
    return ns


def default_render_config(kind: str = "head") -> RenderConfig:
    """The RenderConfig of a Network constructed WITHOUT `args=` (the reference's unchanged constructor call): the
    process's flags.  With no flags anywhere (no `parse_args()` call, no `--config` on the command line) the paper
    model's dimensions are used and said so once -- upstream's bare defaults (`dim_aud = dim_expr = 0`) build a network
    its own forward cannot run."""
    ns = current_args(kind)
    if ns is not None:
        return to_render_config(ns)
    if kind not in _warned:
        _warned.add(kind)
        logger.warning("idealnerf_amd: Network built without flags (no config_parser().parse_args(), no --config on the command "
                       "line, no args=): using the paper model's dimensions (dim_aud 64, dim_expr %s, near 0.3, far 0.9, perturb 1)",
                       "76" if kind == "head" else "79")
    return RenderConfig() if kind == "head" else RenderConfig(dim_expr=79)


def check_against_current(cfg: RenderConfig, kind: str = "head"):
    """A Network given an explicit `args=` while the process's flags name OTHER conditioning widths is a mistake that
    would otherwise surface as a shape error deep inside a render (or not at all): refuse it by name."""
    ns = _current.get(kind)
    if ns is None:
        return
    want = to_render_config(ns)
    bad = [f"{k}: config {getattr(want, k)} != network {getattr(cfg, k)}" for k in ("dim_aud", "dim_expr")
           if (k != "dim_expr" or kind == "head") and getattr(want, k) != getattr(cfg, k)]
    if bad:
        raise ValueError("the parsed config (" + str(getattr(ns, "config", None)) + ") and the RenderConfig passed as args= "
                         "disagree on " + "; ".join(bad) + " -- build the Network without args= to take the config's, or "
                         "config.set_current_args(None) to drop it")


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
