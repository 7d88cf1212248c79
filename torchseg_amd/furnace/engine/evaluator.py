"""Evaluator: the sliding-window / multi-scale / flip inference driver of furnace/engine/evaluator.py, GPU-resident.

Same class surface as the reference (`Evaluator(dataset, class_num, image_mean, image_std, network, multi_scales, is_flip,
devices, verbose, save_path, show_image)`, `run`, `whole_eval`, `sliding_eval`, `scale_process`, `val_func_process`,
`process_image`, subclass hooks `func_per_iteration` / `compute_metric`), so an unchanged eval.py subclasses it as before.
What changed is where the pixels live: the reference resizes every scale with cv2 on the host, normalises and pads in
numpy, uploads each window, downloads each score map and resizes it with cv2 again (evaluator.py:192-193, :250-252,
:286-298).  Here the uint8 image is uploaded once; every network input (scale + normalise + pad + window) is sampled
straight from it by `tsg_augment_crop`, the window scores accumulate on the device, `tsg_resize_bilinear_hp` brings each
scale to the original size and sums the scales, and the confusion matrix can be taken from the summed scores with the
arg-max fused (`hist_from_scores` -> `tsg_confusion_logits`).  Only the final class map leaves the GPU.
"""
import os
import time

import numpy as np
import torch

from engine.logger import get_logger
from utils.pyt_utils import ensure_dir, link_file, load_model

logger = get_logger()


def _cv_round(v):
    return int(round(v))                          # cvRound: half to even


class Evaluator(object):
    window_batch = 4                              # windows per network call (eval-mode BN: results do not depend on it)

    def __init__(self, dataset, class_num, image_mean, image_std, network, multi_scales, is_flip, devices,
                 verbose=False, save_path=None, show_image=False):
        self.dataset = dataset
        self.ndata = self.dataset.get_length() if dataset is not None else 0
        self.class_num = class_num
        self.image_mean = np.asarray(image_mean, dtype=np.float32)
        self.image_std = np.asarray(image_std, dtype=np.float32)
        self.multi_scales = multi_scales
        self.is_flip = is_flip
        self.network = network
        self.devices = devices
        self.val_func = None
        self.verbose = verbose
        self.save_path = save_path
        if save_path is not None:
            ensure_dir(save_path)
        self.show_image = show_image

    # ---- model selection / bookkeeping (evaluator.py:43-94) --------------------------------------------------
    def run(self, model_path, model_indice, log_file, log_file_link):
        """-e *.pth | epoch | start-end | start-  (the four modes of the reference)."""
        if '.pth' in model_indice:
            models = [model_indice]
        elif "-" in model_indice:
            lo, hi = model_indice.split("-")[0], model_indice.split("-")[1]
            lo = int(lo)
            names = [m for m in os.listdir(model_path) if m != "epoch-last.pth"]
            epochs = np.array([int(m.split(".")[0].split("-")[1]) for m in names])
            keep = epochs >= lo
            if hi:
                assert lo < int(hi)
                keep &= epochs <= int(hi)
            models = [os.path.join(model_path, m) for m, k in zip(names, keep) if k]
        else:
            models = [os.path.join(model_path, 'epoch-%s.pth' % model_indice)]
        with open(log_file, 'a') as results:
            link_file(log_file, log_file_link)
            for model in models:
                logger.info("Load Model: %s" % model)
                self.val_func = load_model(self.network, model)
                line = self.multi_process_evaluation()
                results.write('Model: ' + model + '\n' + line + '\n')
                results.flush()

    def multi_process_evaluation(self):
        """One process: the images of every device's share run back to back on devices[0] (the reference forks one
        worker per GPU, evaluator.py:96-151; the per-image work is identical)."""
        start = time.perf_counter()
        device = self.devices[0] if self.devices else 0
        all_results = []
        for idx in range(self.ndata):
            all_results.append(self.func_per_iteration(self.dataset[idx], device))
            if self.verbose:
                self.compute_metric(all_results)
        line = self.compute_metric(all_results)
        logger.info('Evaluation Elapsed Time: %.2fs' % (time.perf_counter() - start))
        return line

    def func_per_iteration(self, data, device):
        raise NotImplementedError

    def compute_metric(self, results):
        raise NotImplementedError

    # ---- device-side pieces ------------------------------------------------------------------------------------
    @staticmethod
    def _device(device):
        return torch.device("cuda", device) if isinstance(device, int) else torch.device(device if device is not None else "cuda")

    def _upload(self, img, device):
        """uint8 HWC image -> device (grey images replicated to 3 channels, evaluator.py:278-282)."""
        if isinstance(img, torch.Tensor):
            t = img
        else:
            t = torch.from_numpy(np.ascontiguousarray(img))
        if t.dim() == 2:
            t = t[:, :, None]
        if t.shape[2] < 3:
            t = t.expand(t.shape[0], t.shape[1], 3)
        return t.to(self._device(device), dtype=torch.uint8).contiguous()

    def _network_scores(self, x):
        """val_func_process (evaluator.py:255-273) on a batch [n,3,h,w]: log-probabilities, flip TTA, exp."""
        # the reference moves the network to the input's device on every call (evaluator.py:258-259: an unchanged eval.py
        # builds it on the CPU and run() loads the checkpoint with map_location='cpu'); a no-op once it is there
        self.val_func.to(x.device)
        self.val_func.eval()
        with torch.no_grad():
            score = self.val_func(x)
            if self.is_flip:
                score = score + self.val_func(x.flip(-1)).flip(-1)
            return torch.exp(score.float())

    def val_func_process(self, input_data, device=None):
        """[3,h,w] normalised input (numpy or tensor) -> exp-scores [C,h,w] on the device."""
        if not isinstance(input_data, torch.Tensor):
            input_data = torch.from_numpy(np.ascontiguousarray(input_data, dtype=np.float32))
        x = input_data.to(self._device(device), dtype=torch.float32)[None]
        return self._network_scores(x)[0]

    def process_image(self, img, crop_size=None):
        """Host form kept for subclasses that call it (evaluator.py:275-298); the sliding path below never does."""
        p = np.asarray(img)
        if p.shape[2] < 3:
            p = np.concatenate((p, p, p), axis=2)
        p = (p.astype(np.float32) / 255.0 - self.image_mean) / self.image_std
        if crop_size is not None:
            if isinstance(crop_size, int):
                crop_size = (crop_size, crop_size)
            ph, pw = max(crop_size[0] - p.shape[0], 0), max(crop_size[1] - p.shape[1], 0)
            margin = np.array([ph // 2, ph // 2 + ph % 2, pw // 2, pw // 2 + pw % 2], np.uint32)
            p = np.pad(p, [(margin[0], margin[1]), (margin[2], margin[3]), (0, 0)], mode="constant")
            return p.transpose(2, 0, 1), margin
        return p.transpose(2, 0, 1)

    def _windows(self, img_d, s, crop_size, stride_rate):
        """Geometry of one scale: (sh, sw, pad_rows, pad_cols, margin, [(s_y, s_x)], raw_pad)."""
        H, W = img_d.shape[0], img_d.shape[1]
        sh, sw = _cv_round(H * s), _cv_round(W * s)                 # dsize of cv2.resize(img, None, fx=s, fy=s)
        long_size = max(sh, sw)
        if long_size <= crop_size:                                  # one window: normalise, THEN pad with 0
            mh, mw = crop_size - sh, crop_size - sw
            margin = (mh // 2, mh // 2 + mh % 2, mw // 2, mw // 2 + mw % 2)
            return sh, sw, crop_size, crop_size, margin, [(0, 0)], False
        stride = int(np.ceil(crop_size * stride_rate))
        pad_rows, pad_cols = max(sh, crop_size), max(sw, crop_size)  # pad_image_to_shape of the RAW image with 0
        mh, mw = pad_rows - sh, pad_cols - sw
        margin = (mh // 2, mh // 2 + mh % 2, mw // 2, mw // 2 + mw % 2)
        r_grid = int(np.ceil((pad_rows - crop_size) / stride)) + 1
        c_grid = int(np.ceil((pad_cols - crop_size) / stride)) + 1
        wins = []
        for gy in range(r_grid):
            for gx in range(c_grid):
                e_x, e_y = min(gx * stride + crop_size, pad_cols), min(gy * stride + crop_size, pad_rows)
                wins.append((e_y - crop_size, e_x - crop_size))
        return sh, sw, pad_rows, pad_cols, margin, wins, True

    def scale_scores(self, img_d, s, crop_size, stride_rate):
        """scale_process (evaluator.py:203-253) up to the final resize: exp-scores [C, sh, sw] of one scale."""
        from torchseg_amd import kernels as K
        kp = K.provider()
        H, W = img_d.shape[0], img_d.shape[1]
        sh, sw, pad_rows, pad_cols, margin, wins, raw_pad = self._windows(img_d, s, crop_size, stride_rate)
        data = None
        for i0 in range(0, len(wins), self.window_batch):
            chunk = wins[i0:i0 + self.window_batch]
            # window origin in PADDED coordinates -> position in the scaled image; a padded dimension has one window at 0,
            # which the kernel centres exactly like pad_image_to_shape
            geom = np.array([[H, W, sh, sw, 0, max(sy - margin[0], 0) if sh >= crop_size else 0,
                              max(sx - margin[2], 0) if sw >= crop_size else 0] for sy, sx in chunk], dtype=np.int32)
            x, _ = kp.augment_crop([img_d] * len(chunk), None, geom, (crop_size, crop_size), self.image_mean, self.image_std,
                                   pad_pixel=0.0 if raw_pad else -1.0, inv_scale=np.full((len(chunk), 2), float(s)))
            t = self._network_scores(x)
            if data is None:
                data = torch.zeros((t.shape[1], pad_rows, pad_cols), dtype=torch.float32, device=t.device)
            for (sy, sx), tt in zip(chunk, t):
                data[:, sy:sy + crop_size, sx:sx + crop_size] += tt
        return data[:, margin[0]:pad_rows - margin[1], margin[2]:pad_cols - margin[3]].contiguous()

    def scale_process(self, img, ori_shape, crop_size, stride_rate, device=None, scale=1.0):
        """Reference signature (img = the already scaled uint8 image): -> float32 numpy [ori_rows, ori_cols, C]."""
        from torchseg_amd import kernels as K
        img_d = self._upload(img, device)
        score = self.scale_scores(img_d, 1.0, crop_size, stride_rate)
        out = K.provider().resize_bilinear_hp(score, ori_shape[0], ori_shape[1])
        return out.permute(1, 2, 0).cpu().numpy()

    def sliding_scores(self, img, crop_size, stride_rate, device=None):
        """Scores summed over the scales at the original size: float32 [C, rows, cols] on the device."""
        from torchseg_amd import kernels as K
        kp = K.provider()
        img_d = self._upload(img, device)
        total = None
        for s in self.multi_scales:
            score = self.scale_scores(img_d, s, crop_size, stride_rate)
            if total is None:
                total = kp.resize_bilinear_hp(score, img_d.shape[0], img_d.shape[1])
            else:
                kp.resize_bilinear_hp(score, img_d.shape[0], img_d.shape[1], out=total, accumulate=True)
        return total

    def sliding_eval(self, img, crop_size, stride_rate, device=None):
        """evaluator.py:186-201 -> class map (numpy int64 [rows, cols])."""
        return self.sliding_scores(img, crop_size, stride_rate, device).argmax(0).cpu().numpy()

    def whole_eval(self, img, output_size, input_size=None, device=None):
        """evaluator.py:162-183: one pass over the whole (optionally padded) image, scores resized to output_size."""
        from torchseg_amd import kernels as K
        img_d = self._upload(img, device)
        H, W = img_d.shape[0], img_d.shape[1]
        if input_size is not None:
            isz = (input_size, input_size) if isinstance(input_size, int) else tuple(input_size)
            geom = np.array([[H, W, H, W, 0, 0, 0]], dtype=np.int32)
            x, _ = K.provider().augment_crop([img_d], None, geom, isz, self.image_mean, self.image_std, pad_pixel=-1.0)
            mh, mw = max(isz[0] - H, 0), max(isz[1] - W, 0)
            pred = self._network_scores(x)[0][:, mh // 2:isz[0] - (mh // 2 + mh % 2), mw // 2:isz[1] - (mw // 2 + mw % 2)]
        else:
            geom = np.array([[H, W, H, W, 0, 0, 0]], dtype=np.int32)
            x, _ = K.provider().augment_crop([img_d], None, geom, (H, W), self.image_mean, self.image_std)
            pred = self._network_scores(x)[0]
        pred = pred.contiguous()
        if output_size is not None:
            pred = K.provider().resize_bilinear_hp(pred, output_size[0], output_size[1])
        return pred.argmax(0).cpu().numpy()

    def hist_from_scores(self, scores, label, out=None):
        """hist_info(class_num, scores.argmax, label) (seg_opr/metric.py:9-20) without materialising the class map:
        int64 [class_num^2 + 3] = confusion matrix, labeled, correct, ignored; accumulates into `out` across images."""
        from torchseg_amd import kernels as K
        lab = torch.as_tensor(np.ascontiguousarray(label)) if not isinstance(label, torch.Tensor) else label
        lab = lab.to(scores.device)
        if lab.dtype not in (torch.int64, torch.uint8):
            lab = lab.to(torch.int64)
        return K.provider().confusion_logits(scores[None].contiguous(), lab[None].contiguous(), self.class_num, out)
