"""`python -m vhap_amd.track` -- the reference's entry script (vhap/track.py:16-21: `cfg = tyro.cli(BaseTrackingConfig);
GlobalTracker(cfg).optimize()`) over this library.

Two ways to hand over the REFERENCE's configuration (its classes are imported from the user's checkout, never re-implemented):

    python -m vhap_amd.track [--checkout /path/to/VHAP] [--nersemble] <tyro arguments of vhap.track ...>
        with `tyro` importable: the reference's own command line, parsed by tyro against the reference's config class
        (vhap.config.base.BaseTrackingConfig, or vhap.config.nersemble.NersembleTrackingConfig with --nersemble)

    python -m vhap_amd.track --config out/.../config.yml [--checkout /path/to/VHAP] [--output-folder DIR] [--device cuda:0]
        the `config.yml` every run of the reference writes into its output folder (vhap/model/tracker.py:1240-1241: yaml.dump(cfg) -- a
        Python-object YAML of the reference's dataclasses; loading it needs nothing but the checkout's config modules)

Then, like the reference: tracker = GlobalTracker.from_reference_config(cfg); tracker.optimize(); the result goes to
<output folder>/<timestamp>/tracked_flame_params.npz (the schema of tracker.py:1152-1218) next to a copy of the config.
Landmarks must exist on disk (`cfg.exp.reuse_landmarks`) or `--landmark-weights <state dict of the face_alignment FAN>` runs the detection
first (vhap_amd.landmarks: the network on the matrix cores; `--face-detector-weights`: the package's `sfd` detector, vhap_amd.face_detector; the STAR detector is not built), and the licensed FLAME assets at
the reference's paths (vhap/model/flame.py:38-46), relative to the working directory."""
import argparse
import os
import sys
import types
from datetime import datetime
from pathlib import Path


def _reference_importable(checkout):
    """`vhap.config.*` importable: the checkout on sys.path; `tyro` stood in for when absent (the config modules import it at the top but
    only USE it under `if __name__ == '__main__'`)."""
    if checkout and checkout not in sys.path:
        sys.path.insert(0, checkout)
    try:
        import tyro  # noqa: F401
    except ImportError:
        m = types.ModuleType("tyro")
        m.cli = lambda *a, **k: (_ for _ in ()).throw(RuntimeError("tyro is not installed: use --config <config.yml>"))
        m.to_yaml = lambda *a, **k: ""
        m.conf, m.extras = types.SimpleNamespace(), types.SimpleNamespace(set_accent_color=lambda *a, **k: None)
        sys.modules["tyro"] = m
    try:
        import vhap.config.base  # noqa: F401
    except ImportError as e:
        raise SystemExit(f"cannot import the reference's config classes ({e}): pass --checkout /path/to/VHAP or put it on PYTHONPATH")


def load_reference_config(path, checkout=None):
    """The reference config object a run of the reference dumped (tracker.py:1240-1241).  The file names Python classes -- the reference's
    dataclasses, pathlib paths -- so it is read with PyYAML's full loader: load only files you wrote."""
    import yaml
    _reference_importable(checkout)
    with open(path, "r", encoding="utf8") as f:
        cfg = yaml.load(f, Loader=yaml.Loader)           # noqa: S506 (python-object YAML by construction)
    if not hasattr(cfg, "pipeline") or not hasattr(cfg, "data"):
        raise SystemExit(f"{path} does not hold a tracking configuration")
    return cfg


def main(argv=None):
    ap = argparse.ArgumentParser(prog="python -m vhap_amd.track", description=__doc__.split("\n\n")[0], add_help=False)
    ap.add_argument("--config", default=None, help="config.yml written by a run of the reference (or of this script)")
    ap.add_argument("--checkout", default=os.environ.get("VHAP_REFERENCE"), help="root of the reference checkout (where vhap/ lives)")
    ap.add_argument("--nersemble", action="store_true", help="tyro mode: parse against NersembleTrackingConfig")
    ap.add_argument("--output-folder", default=None, help="overrides cfg.exp.output_folder")
    ap.add_argument("--device", default=None, help="overrides cfg.device")
    ap.add_argument("--batch-size", type=int, default=None)
    ap.add_argument("--no-evaluate", action="store_true")
    ap.add_argument("--device-prepare", action="store_true", help="colour correction / scale factor / compositing on the device (bit-identical)")
    ap.add_argument("--landmark-weights", default=None, help="state dict of the face_alignment package's FAN: run landmark detection first "
                    "(tracker.py:1263-1277; landmark_source 'face-alignment'), on the matrix cores (vhap_amd.landmarks)")
    ap.add_argument("--face-detector-weights", default=None, help="state dict of the package's `sfd` face detector (S3FD on the matrix cores, "
                    "vhap_amd.face_detector); without it the whole frame is the face box")
    ap.add_argument("--dry-run", action="store_true", help="load + convert the configuration, print the stage plan, do not open data or fit")
    ap.add_argument("-h", "--help", action="store_true")
    a, rest = ap.parse_known_args(argv)
    if a.help and a.config is None and not rest:
        ap.print_help()
        return 0
    if a.config is not None:
        if rest:
            raise SystemExit(f"unknown arguments with --config: {rest}")
        cfg = load_reference_config(a.config, a.checkout)
    else:
        _reference_importable(a.checkout)
        import tyro
        if a.nersemble:
            from vhap.config.nersemble import NersembleTrackingConfig as Cfg
        else:
            from vhap.config.base import BaseTrackingConfig as Cfg
        cfg = tyro.cli(Cfg, args=rest + (["--help"] if a.help else []))
    if a.output_folder is not None:
        cfg.exp.output_folder = Path(a.output_folder)
    if a.device is not None:
        cfg.device = a.device
    from .reference_adapter import convert_config
    mine = convert_config(cfg)
    stages = ["lmk_init_rigid", "lmk_init_all"] + (["rgb_init_texture", "rgb_init_all"] + (["rgb_init_offset"] if mine.model.use_static_offset else [])
                                                   if mine.exp.photometric else [])
    stages += ["rgb_sequential_tracking" if mine.exp.photometric else "lmk_sequential_tracking",
               "rgb_global_tracking" if mine.exp.photometric else "lmk_global_tracking"]
    print(f"[vhap_amd.track] sequence {getattr(cfg.data, 'sequence', '?')} under {getattr(cfg.data, 'root_folder', '?')}; device {mine.device}; "
          f"stages: {', '.join(stages)}", flush=True)
    if a.dry_run:
        return 0
    if a.landmark_weights is not None:
        from .reference_adapter import detect_landmarks
        detect_landmarks(cfg, a.landmark_weights, checkout=a.checkout, device=mine.device, face_detector_weights=a.face_detector_weights)
    from .tracker import GlobalTracker
    tracker = GlobalTracker.from_reference_config(cfg, checkout=a.checkout, device_prepare=a.device_prepare)
    out_dir = Path(cfg.exp.output_folder) / datetime.now().strftime("%Y-%m-%d_%H-%M-%S")      # tracker.py:1231-1232
    out_dir.mkdir(parents=True, exist_ok=True)
    import yaml
    (out_dir / "config.yml").write_text(yaml.dump(cfg), "utf8")                               # tracker.py:1240-1241
    report = tracker.optimize(batch_size=a.batch_size, evaluate=not a.no_evaluate)
    tracker.save_result(str(out_dir / "tracked_flame_params.npz"))
    print(f"[vhap_amd.track] {tracker.n_timesteps} timesteps fitted -> {out_dir / 'tracked_flame_params.npz'}" +
          (f"; evaluation: {report}" if report is not None else ""), flush=True)
    return 0


if __name__ == "__main__":
    sys.exit(main())
