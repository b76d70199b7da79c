"""Drop-in for BodySLAM_not_refactored/MDEM/mdem_interface.py (MDEMInterface, :17-83): legacy names of the
same MDEM API.  ``save_depth_map`` APPENDS the extension (FrameIO.save_p_img, UTILS/io_utils.py:49-74)."""
import warnings

import numpy as np
import torch
from PIL import Image

from .weights import load_zoedepth_weights
from .zoedepth import ZOED_K, ZOED_N, ZOED_NK, ZoeDepthEngine


class MDEMInterface:
    def __init__(self, model_type: str = "ZoeD_NK", weights=None, dtype=torch.float16, precision: str = "accurate"):
        self.zoe = self._initialize_ZOE(model_type, weights, dtype, precision)

    def _initialize_ZOE(self, model_type: str, weights=None, dtype=torch.float16, precision: str = "accurate"):
        if model_type not in ("ZoeD_N", "ZoeD_K", "ZoeD_NK"):
            # mdem_interface.py:42-44 warns (and then still asks the hub for the bad name); here: warn + default
            warnings.warn(f"The model type selected [{model_type}], does not exist! Using default model [ZoeD_NK]")
            model_type = "ZoeD_NK"
        sd = weights if isinstance(weights, dict) else load_zoedepth_weights(weights)
        print("[INFO] model loaded on cuda (MI355X, HIP)")
        cfg = {"ZoeD_NK": ZOED_NK, "ZoeD_N": ZOED_N, "ZoeD_K": ZOED_K}[model_type]      # one / two metric heads
        return ZoeDepthEngine(sd, cfg, dtype=dtype, precision=precision)

    def infer_monocular_depth_map(self, path_to_frame: str) -> Image.Image:
        image = Image.open(path_to_frame).convert("RGB")
        frame = torch.from_numpy(np.asarray(image, dtype=np.uint8).copy()).unsqueeze(0).cuda()
        _, u16 = self.zoe.infer(frame, flip_aug=True)
        return Image.fromarray(u16[0].cpu().numpy().view(np.uint16))   # mode "I;16"

    @staticmethod
    def save_depth_map(image: Image.Image, saving_path: str, extension: str = None):
        try:
            if extension is None:
                warnings.warn("No extension has been provided")
                image.save(saving_path)
                return True
            image.save(saving_path + extension)
            return True
        except ValueError as e:
            print(f"Error while saving: {e}")

    def debug(self, path_to_frame: str, saving_path: str):
        """mdem_interface.py:85-121: exercise the inference and the saving method and report each.  (The reference's version calls both
        unbound and cannot run; this one does what its docstring says and returns the list of outcomes.)"""
        passed = []
        for name, fn in (("infer method", lambda: self.infer_monocular_depth_map(path_to_frame)),
                         ("saving method", lambda: self.save_depth_map(self.infer_monocular_depth_map(path_to_frame), saving_path, ".png"))):
            print(f"[DEBUG]: Testing {name}...")
            try:
                fn()
                passed.append(True)
                print(f"[DEBUG]: {name} status -> ok")
            except Exception as e:
                passed.append(False)
                print(f"[DEBUG]: OPS :/ -> {e}")
        return passed
