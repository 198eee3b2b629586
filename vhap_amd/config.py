"""Tracking configuration -- the API surface of vhap/config/base.py (and nersemble.py) as plain
dataclasses (the reference parses the same tree with tyro, which is not a dependency of the hot path).

Field names, defaults and the `cfg.pipeline[stage]` / `cfg.w['reg_neck']` item access
(base.py:24-30) are kept so that code written against the reference config reads unchanged.  Only the
fields the hot path consumes are present; `render.backend` gains the value 'hip'.
"""
from dataclasses import dataclass, field
from typing import Optional, Tuple


class Config:
    def __getitem__(self, name: str):                                   # base.py:25-30
        if hasattr(self, name):
            return getattr(self, name)
        raise AttributeError(f"{self.__class__.__name__} has no attribute '{name}'")


@dataclass
class DataConfig(Config):                                               # base.py:33-57 (subset)
    calibrated: bool = False
    n_downsample_rgb: Optional[int] = None
    scale_factor: float = 1.0
    background_color: Optional[str] = "white"


@dataclass
class ModelConfig(Config):                                              # base.py:60-91
    n_shape: int = 300
    n_expr: int = 100
    n_tex: int = 100
    use_static_offset: bool = True
    use_dynamic_offset: bool = False
    add_teeth: bool = True
    remove_lip_inside: bool = False
    tex_resolution: int = 2048
    tex_painted: bool = True
    tex_extra: bool = True
    tex_clusters: Tuple[str, ...] = ("skin", "hair", "boundary", "lips_tight", "teeth", "sclerae", "irises")
    residual_tex: bool = True
    occluded: Tuple[str, ...] = ()


@dataclass
class RenderConfig(Config):                                             # base.py:93-110
    backend: str = "hip"
    use_opengl: bool = False
    background_train: str = "target"
    disturb_rate_fg: Optional[float] = 0.5
    disturb_rate_bg: Optional[float] = 0.5
    background_eval: str = "target"
    lighting_type: str = "SH"
    lighting_space: str = "world"


@dataclass
class LearningRateConfig(Config):                                       # base.py:113-122
    base: float = 5e-3
    translation: float = 1e-3
    expr: float = 5e-2
    static_offset: float = 5e-4
    dynamic_offset: float = 5e-4
    camera: float = 5e-3
    light: float = 5e-3


@dataclass
class LossWeightConfig(Config):                                         # base.py:125-196
    landmark: Optional[float] = 10.0
    always_enable_jawline_landmarks: bool = True
    photo: Optional[float] = 30.0
    reg_shape: float = 3e-1
    reg_neck: float = 3e-1
    reg_jaw: float = 3e-1
    reg_eyes: float = 3e-2
    reg_expr: float = 3e-2
    reg_tex_res_clusters: Optional[float] = 1e1
    reg_tex_res_for: Tuple[str, ...] = ("sclerae", "teeth")
    reg_tex_tv: Optional[float] = 1e4
    reg_tex_pca: float = 1e-4
    reg_light: Optional[float] = None
    reg_diffuse: Optional[float] = 1e2
    reg_offset: Optional[float] = 3e2
    reg_offset_relax_coef: float = 1.0
    reg_offset_relax_for: Tuple[str, ...] = ("hair", "ears")
    reg_offset_lap: Optional[float] = 1e6
    reg_offset_lap_relax_coef: float = 0.1
    reg_offset_lap_relax_for: Tuple[str, ...] = ("hair", "ears")
    reg_offset_rigid: Optional[float] = 3e2
    reg_offset_rigid_for: Tuple[str, ...] = ("left_ear", "right_ear", "neck", "left_eye", "right_eye", "lips_tight")
    reg_offset_dynamic: Optional[float] = 3e5
    blur_iter: int = 0
    smooth_trans: float = 3e2
    smooth_rot: float = 3e1
    smooth_neck: float = 3e1
    smooth_jaw: float = 1e-1
    smooth_eyes: float = 0
    smooth_expr: float = 1e0


@dataclass
class StageConfig(Config):                                              # base.py:221-224
    disable_jawline_landmarks: bool = False


@dataclass
class StageLmkInitRigidConfig(StageConfig):
    num_steps: int = 500
    optimizable_params: Tuple[str, ...] = ("cam", "pose")


@dataclass
class StageLmkInitAllConfig(StageConfig):
    num_steps: int = 500
    optimizable_params: Tuple[str, ...] = ("cam", "pose", "shape", "joints", "expr")


@dataclass
class StageLmkSequentialTrackingConfig(StageConfig):
    num_steps: int = 50
    optimizable_params: Tuple[str, ...] = ("pose", "joints", "expr")


@dataclass
class StageLmkGlobalTrackingConfig(StageConfig):
    num_epochs: int = 30
    optimizable_params: Tuple[str, ...] = ("cam", "pose", "shape", "joints", "expr")


@dataclass
class PhotometricStageConfig(StageConfig):                              # base.py:250-255
    align_texture_except: Tuple[str, ...] = ()
    align_boundary_except: Tuple[str, ...] = ("bottomline",)


@dataclass
class StageRgbInitTextureConfig(PhotometricStageConfig):
    num_steps: int = 500
    optimizable_params: Tuple[str, ...] = ("cam", "shape", "texture", "lights")
    align_texture_except: Tuple[str, ...] = ("hair", "boundary", "neck")
    align_boundary_except: Tuple[str, ...] = ("hair", "boundary")


@dataclass
class StageRgbInitAllConfig(PhotometricStageConfig):
    num_steps: int = 500
    optimizable_params: Tuple[str, ...] = ("cam", "pose", "shape", "joints", "expr", "texture", "lights")
    disable_jawline_landmarks: bool = True
    align_texture_except: Tuple[str, ...] = ("hair", "boundary", "neck")
    align_boundary_except: Tuple[str, ...] = ("hair", "bottomline")


@dataclass
class StageRgbInitOffsetConfig(PhotometricStageConfig):
    num_steps: int = 500
    optimizable_params: Tuple[str, ...] = ("cam", "pose", "shape", "joints", "expr", "texture", "lights", "static_offset")
    disable_jawline_landmarks: bool = True
    align_texture_except: Tuple[str, ...] = ("hair", "boundary", "neck")


@dataclass
class StageRgbSequentialTrackingConfig(PhotometricStageConfig):
    num_steps: int = 50
    optimizable_params: Tuple[str, ...] = ("pose", "joints", "expr", "texture", "dynamic_offset")
    disable_jawline_landmarks: bool = True


@dataclass
class StageRgbGlobalTrackingConfig(PhotometricStageConfig):
    num_epochs: int = 30
    optimizable_params: Tuple[str, ...] = ("cam", "pose", "shape", "joints", "expr", "texture", "lights",
                                           "static_offset", "dynamic_offset")
    disable_jawline_landmarks: bool = True


@dataclass
class PipelineConfig(Config):                                           # base.py:296-307
    lmk_init_rigid: StageLmkInitRigidConfig = field(default_factory=StageLmkInitRigidConfig)
    lmk_init_all: StageLmkInitAllConfig = field(default_factory=StageLmkInitAllConfig)
    lmk_sequential_tracking: StageLmkSequentialTrackingConfig = field(default_factory=StageLmkSequentialTrackingConfig)
    lmk_global_tracking: StageLmkGlobalTrackingConfig = field(default_factory=StageLmkGlobalTrackingConfig)
    rgb_init_texture: StageRgbInitTextureConfig = field(default_factory=StageRgbInitTextureConfig)
    rgb_init_all: StageRgbInitAllConfig = field(default_factory=StageRgbInitAllConfig)
    rgb_init_offset: StageRgbInitOffsetConfig = field(default_factory=StageRgbInitOffsetConfig)
    rgb_sequential_tracking: StageRgbSequentialTrackingConfig = field(default_factory=StageRgbSequentialTrackingConfig)
    rgb_global_tracking: StageRgbGlobalTrackingConfig = field(default_factory=StageRgbGlobalTrackingConfig)


@dataclass
class ExperimentConfig(Config):
    photometric: bool = True


@dataclass
class BaseTrackingConfig(Config):                                       # base.py:310-358
    data: DataConfig = field(default_factory=DataConfig)
    model: ModelConfig = field(default_factory=ModelConfig)
    render: RenderConfig = field(default_factory=RenderConfig)
    exp: ExperimentConfig = field(default_factory=ExperimentConfig)
    lr: LearningRateConfig = field(default_factory=LearningRateConfig)
    w: LossWeightConfig = field(default_factory=LossWeightConfig)
    pipeline: PipelineConfig = field(default_factory=PipelineConfig)
    batch_size: Optional[int] = 16
    device: str = "cuda"

    def __post_init__(self):
        if not self.model.use_static_offset and not self.model.use_dynamic_offset:
            self.model.occluded = tuple(list(self.model.occluded) + ["hair"])
        for st in self.pipeline.__dict__.values():
            if isinstance(st, PhotometricStageConfig):
                st.align_texture_except = tuple(list(st.align_texture_except) + list(self.model.occluded))
                st.align_boundary_except = tuple(list(st.align_boundary_except) + list(self.model.occluded))


def nersemble_config(**kw):
    """The overrides of vhap/config/nersemble.py:22-60 (calibrated multi-view capture)."""
    cfg = BaseTrackingConfig(**kw)
    cfg.data.calibrated = True
    cfg.data.background_color = None
    cfg.w.landmark = 3.0
    cfg.w.always_enable_jawline_landmarks = False
    cfg.w.reg_expr = 1e-2
    cfg.w.reg_tex_tv = 1e5
    cfg.w.smooth_expr = 0
    cfg.pipeline.rgb_sequential_tracking.optimizable_params = ("pose", "joints", "expr", "dynamic_offset")
    for st in (cfg.pipeline.rgb_sequential_tracking, cfg.pipeline.rgb_global_tracking):
        st.align_texture_except = tuple(["boundary"] + list(cfg.model.occluded))
        st.align_boundary_except = tuple(["boundary"] + list(cfg.model.occluded))
    return cfg
