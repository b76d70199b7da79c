"""Drop-in for BodySLAM_not_refactored/MPEM/mpem_interface.py (MPEMInterface, :23-99)."""
import numpy as np
import torch
from PIL import Image

from .cyclepose import CyclePoseEngine
from .weights import load_cyclepose_checkpoint


class MPEMInterface:
    def __init__(self, path_to_model, dtype=torch.float16, precision: str = "accurate"):
        """``path_to_model``: the reference's checkpoint file (ModelIO container) or a loaded state dict."""
        self.input_shape = (6, 256, 256)
        self.device = "cuda"
        self.pose_model = self._initialize_pose_model(path_to_model, dtype, precision)

    def _initialize_pose_model(self, path_to_model, dtype=torch.float16, precision: str = "accurate"):
        print(f"[INFO] model loaded on {self.device}")
        sd = path_to_model if isinstance(path_to_model, dict) else load_cyclepose_checkpoint(path_to_model)
        return CyclePoseEngine(sd, dtype=dtype, precision=precision)

    def infer_relative_pose_between(self, path_frame1, path_frame2, type_of_trans='crop'):
        """two frame paths -> (4,4) float32 SE(3) relative pose (prev -> curr)."""
        assert type_of_trans == 'crop' or type_of_trans == 'resize', "type_of_trans must be 'crop' or 'resize'!"
        if type_of_trans == 'resize':
            # transforms.Resize(128) keeps the aspect ratio (128x170 for 640x480): the reference's lazily sized
            # skip_linear then has a different, never-trained shape (architecture_v3.py:205-209)
            raise NotImplementedError("type_of_trans='resize' is not built: no checkpoint defines skip_linear for it")
        f1 = np.asarray(Image.open(path_frame1), dtype=np.uint8)
        f2 = np.asarray(Image.open(path_frame2), dtype=np.uint8)
        if f1.ndim != 3 or f1.shape[2] != 3 or f1.shape != f2.shape:
            raise ValueError(f"expected two RGB frames of equal size, got {f1.shape} and {f2.shape}")
        frames = torch.from_numpy(np.stack([f1, f2])).cuda()
        pairs = torch.tensor([[0, 1]], dtype=torch.int32, device="cuda")
        T = self.pose_model.infer_pairs(frames, pairs)
        return T[0].cpu().numpy()
