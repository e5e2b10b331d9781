"""`python inference.py --cfg configs/t2v_infer.yaml [--debug] [key value ...]` — same entry as the reference's
inference.py:16-18, dispatching on TASK_TYPE through the INFER_ENGINE registry, over the MI355X-native hot path."""
from videomv_amd.config import Config
from videomv_amd.registry import INFER_ENGINE
import videomv_amd  # noqa: F401  (registers MODEL / DIFFUSION / AUTO_ENCODER / INFER_ENGINE plugins)
import videomv_amd.entrance  # noqa: F401

if __name__ == '__main__':
    cfg_update = Config(load=True)
    INFER_ENGINE.build(dict(type=cfg_update.TASK_TYPE), cfg_update=cfg_update.cfg_dict)
