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
        # FrameIO.load_p_img(path) (io_utils.py:33-47) opens with convert_to_rgb=False, i.e. WITHOUT a colour conversion: for the RGB JPEG / PNG
        # frames of the reference's datasets the two are the same image.  A greyscale or RGBA file would reach the reference's network as a
        # 1- or 4-channel tensor and fail in its first convolution (6 input channels = two RGB frames); here it is converted to RGB instead --
        # a deliberate divergence on inputs the reference cannot process at all.
        i1, i2 = Image.open(path_frame1).convert('RGB'), Image.open(path_frame2).convert('RGB')
        window = None
        if type_of_trans == 'resize':
            # transforms.Resize(128) (mpem_interface.py:45-50): the smaller edge becomes 128, aspect kept, PIL bilinear -- the same
            # PIL call torchvision makes, on the host as in the reference; the whole resized frame is the network input
            i1, i2 = self._resize128(i1), self._resize128(i2)
            window = "full"
        f1, f2 = np.asarray(i1, dtype=np.uint8), np.asarray(i2, dtype=np.uint8)
        if f1.ndim != 3 or f1.shape[2] != 3 or f1.shape != f2.shape:
            raise ValueError(f"expected two RGB frames of equal size, got {f1.shape} and {f2.shape}")
        if window is not None:
            self._ensure_skip(f1.shape[0], f1.shape[1])
        frames = torch.from_numpy(np.stack([f1, f2])).cuda()
        pairs = torch.tensor([[0, 1]], dtype=torch.int32, device="cuda")
        T = self.pose_model.infer_pairs(frames, pairs, window=window)
        return T[0].cpu().numpy()

    @staticmethod
    def _resize128(img: Image.Image, size: int = 128) -> Image.Image:
        """torchvision.transforms.Resize(size) on a PIL image: smaller edge -> size, the other int(size * long / short)."""
        w, h = img.size
        if (w <= h and w == size) or (h <= w and h == size):
            return img
        if w < h:
            ow, oh = size, int(size * h / w)
        else:
            oh, ow = size, int(size * w / h)
        return img.resize((ow, oh), Image.BILINEAR)

    def _ensure_skip(self, h: int, w: int):
        """The reference builds skip_linear lazily from the first input it sees (architecture_v3.py:205-209), i.e. for a resized
        input it is a freshly initialised nn.Linear that no checkpoint holds.  The same is done here -- a default-initialised
        Linear(512 + 256*h'*w', 7), with a warning -- unless weights for that size were registered (CyclePoseEngine.add_skip)."""
        eng = self.pose_model
        h2, w2 = eng.map_hw(h, w)
        if h2 * w2 in eng._skip:
            return
        import warnings
        warnings.warn(f"type_of_trans='resize': no skip_linear weights for a {h}x{w} input; using a randomly initialised layer "
                      "as the reference does (architecture_v3.py:208-209)")
        lin = torch.nn.Linear(512 + 256 * h2 * w2, 7)
        eng.add_skip(lin.weight.data, lin.bias.data)
