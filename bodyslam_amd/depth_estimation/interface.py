"""Drop-in for BodySLAM_Refactored/src/depth_estimation/interface.py (DepthEstimator, :16-107):
same class name, attributes, method names, argument meaning, return types and error behaviour; the
model behind ``.model`` is the HIP ZoeDepth engine instead of a torch.hub module.
"""
import os
import warnings
from typing import Optional

import numpy as np
import torch
from PIL import Image

from ..weights import load_zoedepth_weights
from ..zoedepth import ZOED_K, ZOED_N, ZOED_NK, ZoeDepthEngine


class DepthEstimator:
    '''A class to interface with ZOE for monocular depth estimation'''

    SUPPORTED_MODELS = ['ZoeD_N', 'ZoeD_K', 'ZoeD_NK']
    DEFAULT_MODEL = 'ZoeD_NK'

    def __init__(self, model_type: str = DEFAULT_MODEL, weights=None, dtype=torch.float16, precision: str = "accurate"):
        """``weights``: a state-dict path, a loaded HF-named state dict, or None (BODYSLAM_ZOEDEPTH_WEIGHTS).
        ``precision``: "accurate" (split-precision products, depth within 1e-4 m of the fp32 reference) or "fast"."""
        self.model = self._initialize_model(model_type, weights, dtype, precision)

    def _initialize_model(self, model_type: str, weights=None, dtype=torch.float16, precision: str = "accurate") -> ZoeDepthEngine:
        if model_type not in self.SUPPORTED_MODELS:
            # interface.py:37-40: warn and fall back to the default
            warnings.warn(
                f"The model type '{model_type}' is not supported. Using default model '{self.DEFAULT_MODEL}'.")
            model_type = self.DEFAULT_MODEL
        sd = weights if isinstance(weights, dict) else load_zoedepth_weights(weights)
        print("[INFO] Model loaded on cuda (MI355X, HIP)")
        cfg = {"ZoeD_NK": ZOED_NK, "ZoeD_N": ZOED_N, "ZoeD_K": ZOED_K}[model_type]      # one / two metric heads
        return ZoeDepthEngine(sd, cfg, dtype=dtype, precision=precision)

    def infer_depth_map(self, path_to_frame: str) -> Image.Image:
        """path -> PIL 'I;16' depth map (metres x 256, as upstream infer_pil(output_type="pil"))."""
        image = self.load_image(path_to_frame)
        frame = torch.from_numpy(np.asarray(image, dtype=np.uint8).copy()).unsqueeze(0).cuda()
        _, u16 = self.model.infer(frame, flip_aug=True)
        arr = u16[0].cpu().numpy().view(np.uint16)
        return Image.fromarray(arr)          # uint16 array -> mode "I;16"

    @staticmethod
    def load_image(path: str) -> Image.Image:
        image = Image.open(path)
        return image.convert('RGB')

    @staticmethod
    def save_depth_map(image: Image.Image, saving_path: str, extension: Optional[str] = None):
        if extension:
            saving_path = os.path.splitext(saving_path)[0] + '.' + extension.lstrip('.')
        image.save(saving_path)

    def debug(self, path_to_frame: str, saving_path: str):
        """interface.py:88-107: run the class's methods once each (load, infer, save) and print how each went; nothing is raised."""
        checks = [("load image", lambda: self.load_image(path_to_frame)),
                  ("infer method", lambda: self.infer_depth_map(path_to_frame)),
                  ("saving method", lambda: self.save_depth_map(Image.new('RGB', (100, 100)), saving_path))]
        for name, fn in checks:
            print(f"[DEBUG]: Testing {name}...")
            try:
                fn()
                print(f"[DEBUG]: {name} status -> ok")
            except Exception as e:
                print(f"[DEBUG]: OPS :/ -> {e}")
