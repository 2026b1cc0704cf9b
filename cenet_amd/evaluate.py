"""Evaluation path on the device: batched inference, argmax masks and Dice without leaving HBM.

Mirrors the reference's call sites:
  * `val()` of src/main_acdc.py:218-231 — per batch `dc(argmax(softmax(net(x))), label)` with medpy's binary Dice (non-zero
    = foreground) averaged over the loader  -> `val_dice`, `validate`
  * `calculate_dice_percase` / `test_single_volume` of src/utils/metrics_eval.py:24-34,37-84 — per foreground class over a
    whole volume, with the wrapper rules (pred>0 & gt==0 -> 1, otherwise 0)  -> `volume_class_dice`
The reference predicts slice by slice with batch size 1 and computes the metric on the host; here a volume's slices go
through the network in batches (under `ops.batch1_semantics()`, so that every slice is computed exactly as a batch-1
forward would: CCU's BatchNorm1d is the one batch-size-dependent op of the eval-mode network, cfam.py:260) and one kernel
produces the masks and the overlap counts (loss_optim.hip).
  * `calculate_metric_percase` of src/utils/metrics_eval.py:9-21 — (dice, hd95, jaccard, assd) per class  ->
    `metric_percase`, `test_single_volume`.  medpy's surface distances (border extraction + distance of every border voxel
    to the other border) run on the device in exact integer arithmetic (metrics.hip); only the final percentile / mean of
    the few thousand distances is taken on the host, in float64 like numpy does.
The scipy cubic `zoom` of slices that do not have the network's input size is the reference's own host-side dependency
(metrics_eval.py:45,55) and is called the same way here.
"""
from __future__ import annotations

from typing import Iterable, List, Sequence, Tuple

import numpy as np
import torch

from . import kern, ops


def predict_counts(logits: torch.Tensor, labels: torch.Tensor = None) -> Tuple[torch.Tensor, torch.Tensor]:
    """logits [B,K,H,W] -> (pred [B,H,W] class ids as float, counts [K+1,3] int32 or None).
    counts[c] = (|pred==c & gt==c|, |pred==c|, |gt==c|); counts[K] = the same for the binary masks pred>0 / gt>0."""
    logits = logits.contiguous()
    B, K = logits.shape[:2]
    HW = logits.numel() // (B * K)
    pred = torch.empty((B,) + tuple(logits.shape[2:]), device=logits.device, dtype=torch.float32)
    counts = None
    if labels is not None:
        labels = labels.reshape(B, -1).contiguous().float()
        counts = torch.empty((K + 1, 3), device=logits.device, dtype=torch.int32)
    kern.argmax_counts(logits, labels, pred, counts, B, K, HW)
    return pred, counts


def binary_dice(counts: torch.Tensor) -> float:
    """medpy.metric.binary.dc on the non-zero masks (main_acdc.py:228): 2|A∩B| / (|A|+|B|), 0.0 when both are empty."""
    inter, npred, ngt = (int(v) for v in counts[-1].tolist())
    return 2.0 * inter / (npred + ngt) if (npred + ngt) else 0.0


def class_dice(counts: torch.Tensor) -> List[float]:
    """calculate_dice_percase (metrics_eval.py:24-34) for classes 1..K-1."""
    out = []
    for inter, npred, ngt in counts[1:-1].tolist():
        if npred > 0 and ngt > 0:
            out.append(2.0 * inter / (npred + ngt))
        elif npred > 0 and ngt == 0:
            out.append(1.0)
        else:
            out.append(0.0)
    return out


@torch.no_grad()
def val_dice(net, images: torch.Tensor, labels: torch.Tensor) -> float:
    """one iteration of val() (main_acdc.py:222-228): images [B,Cin,H,W], labels [B,H,W] (or [B,1,H,W])."""
    _, counts = predict_counts(net(images), labels)
    return binary_dice(counts)


@torch.no_grad()
def validate(net, batches: Iterable[Tuple[torch.Tensor, torch.Tensor]]) -> float:
    """val() of main_acdc.py:218-231: mean over the loader of the per-batch binary Dice; sets eval mode like the reference."""
    net.eval()
    vals = [val_dice(net, x, y) for x, y in batches]
    return sum(vals) / max(len(vals), 1)


@torch.no_grad()
def volume_class_dice(net, volume: torch.Tensor, label: torch.Tensor, classes: int, batch_slices: int = 32) -> List[float]:
    """test_single_volume (metrics_eval.py:37-84) for slices that already have the network input size: volume [D,H,W],
    label [D,H,W]; returns the Dice of classes 1..classes-1 over the whole volume."""
    net.eval()
    total = torch.zeros((classes + 1, 3), device=volume.device, dtype=torch.int64)
    for s in range(0, volume.shape[0], batch_slices):
        x = volume[s:s + batch_slices].unsqueeze(1).float()
        with ops.batch1_semantics():  # the reference feeds one slice per forward (metrics_eval.py:46-49): CCU skips its BN
            logits = net(x)
        _, counts = predict_counts(logits, label[s:s + batch_slices])
        total += counts.long()
    return class_dice(total)


# ---------------------------------------------------------------------------------------------------------------------
# surface-distance metrics (metrics_eval.py:9-21 -> medpy.metric.binary.{hd95, assd, jc})
# ---------------------------------------------------------------------------------------------------------------------
def surface_points(mask: torch.Tensor) -> torch.Tensor:
    """Border voxels of a boolean volume [D,H,W] (or image [H,W]) as int32 (z,y,x) rows: mask XOR its erosion by the
    face-connected structuring element with background outside the array (medpy `__surface_distances`)."""
    if mask.dim() == 2:
        mask = mask.unsqueeze(0)
        # a 2-D input has no z neighbours in medpy (2-D structuring element): pad so the z faces count as set
        m = (mask != 0).to(torch.uint8).repeat(3, 1, 1).contiguous()
        border = torch.empty_like(m)
        kern.surface_border(m, border, 3, m.shape[1], m.shape[2])
        pts = torch.nonzero(border[1:2]).to(torch.int32).contiguous()
        return pts
    m = (mask != 0).to(torch.uint8).contiguous()
    border = torch.empty_like(m)
    kern.surface_border(m, border, m.shape[0], m.shape[1], m.shape[2])
    return torch.nonzero(border).to(torch.int32).contiguous()


def _directed_sq(a_pts: torch.Tensor, b_pts: torch.Tensor) -> torch.Tensor:
    out = torch.full((a_pts.shape[0],), 2 ** 31 - 1, dtype=torch.int32, device=a_pts.device)
    kern.min_sqdist(a_pts, b_pts, out)
    return out


def surface_distances(result: torch.Tensor, reference: torch.Tensor) -> Tuple[np.ndarray, np.ndarray]:
    """Both directed surface-distance sets (result->reference, reference->result) as float64 arrays, unit voxel spacing.
    Raises RuntimeError on an empty mask, as medpy does."""
    ra, rb = surface_points(result), surface_points(reference)
    if ra.shape[0] == 0:
        raise RuntimeError("The first supplied array does not contain any binary object.")
    if rb.shape[0] == 0:
        raise RuntimeError("The second supplied array does not contain any binary object.")
    d1 = _directed_sq(ra, rb).cpu().numpy().astype(np.float64)
    d2 = _directed_sq(rb, ra).cpu().numpy().astype(np.float64)
    return np.sqrt(d1), np.sqrt(d2)


def hd95(result: torch.Tensor, reference: torch.Tensor) -> float:
    d1, d2 = surface_distances(result, reference)
    return float(np.percentile(np.hstack((d1, d2)), 95))


def assd(result: torch.Tensor, reference: torch.Tensor) -> float:
    d1, d2 = surface_distances(result, reference)
    return float(np.mean((d1.mean(), d2.mean())))


def metric_percase(pred: torch.Tensor, gt: torch.Tensor) -> Tuple[float, float, float, float]:
    """calculate_metric_percase (metrics_eval.py:9-21): (dice, hd95, jaccard, assd) of two binary volumes, with its rules
    for empty masks."""
    p, g = pred != 0, gt != 0
    np_, ng = int(p.sum()), int(g.sum())
    if np_ > 0 and ng > 0:
        inter = int((p & g).sum())
        dice = 2.0 * inter / float(np_ + ng)
        jac = inter / float(np_ + ng - inter)
        d1, d2 = surface_distances(p, g)
        return dice, float(np.percentile(np.hstack((d1, d2)), 95)), jac, float(np.mean((d1.mean(), d2.mean())))
    if np_ > 0 and ng == 0:
        return 1, 0, 1, 0
    return 0, 0, 0, 0


@torch.no_grad()
def test_single_volume(image, label, net, classes: int, patch_size: Sequence[int] = (224, 224), batch_slices: int = 32,
                       device=None) -> List[Tuple[float, float, float, float]]:
    """test_single_volume (metrics_eval.py:37-84) for a 3-D volume: image / label [1,D,H,W] or [D,H,W] (tensor or ndarray).
    Slices are resized to `patch_size` with scipy's cubic zoom and the predictions back with order 0 exactly as the
    reference does (host side), but the network runs on batches of slices and the metrics on the device.
    Returns [(dice, hd95, jaccard, assd)] for classes 1..classes-1."""
    from scipy.ndimage import zoom
    image = image.squeeze(0).cpu().numpy() if torch.is_tensor(image) else np.asarray(image)
    label = label.squeeze(0).cpu().numpy() if torch.is_tensor(label) else np.asarray(label)
    assert image.ndim == 3, "volume [D,H,W] expected"
    device = device or next(net.parameters()).device
    net.eval()
    D, x, y = image.shape
    resize = (x != patch_size[0] or y != patch_size[1])
    prediction = np.zeros_like(label)
    for s in range(0, D, batch_slices):
        sl = image[s:s + batch_slices]
        if resize:
            sl = np.stack([zoom(a, (patch_size[0] / x, patch_size[1] / y), order=3) for a in sl])
        inp = torch.from_numpy(np.ascontiguousarray(sl)).unsqueeze(1).float().to(device)
        with ops.batch1_semantics():  # slice-by-slice semantics of the reference (CCU's `if B > 1` BatchNorm, cfam.py:260)
            logits = net(inp)
        pred, _ = predict_counts(logits)
        out = pred.reshape(-1, patch_size[0], patch_size[1]).cpu().numpy()
        for i in range(out.shape[0]):
            prediction[s + i] = zoom(out[i], (x / patch_size[0], y / patch_size[1]), order=0) if resize else out[i]
    pred_d = torch.from_numpy(np.ascontiguousarray(prediction)).to(device)
    lab_d = torch.from_numpy(np.ascontiguousarray(label)).to(device)
    return [metric_percase(pred_d == c, lab_d == c) for c in range(1, classes)]


# ---------------------------------------------------------------------------------------------------------------------------
# Synapse (utils/utils_synapse.py:12-100) and skin (utils/utils_skin.py:97-170) evaluation wrappers
# ---------------------------------------------------------------------------------------------------------------------------
def synapse_metric_percase(pred: torch.Tensor, gt: torch.Tensor) -> Tuple[float, float]:
    """calculate_metric_percase of utils_synapse.py:12-21: (dice, hd95) of two binary volumes; (1, 0) for a prediction without
    ground truth, (0, 0) otherwise."""
    p, g = pred != 0, gt != 0
    np_, ng = int(p.sum()), int(g.sum())
    if np_ > 0 and ng > 0:
        inter = int((p & g).sum())
        d1, d2 = surface_distances(p, g)
        return 2.0 * inter / float(np_ + ng), float(np.percentile(np.hstack((d1, d2)), 95))
    if np_ > 0 and ng == 0:
        return 1, 0
    return 0, 0


@torch.no_grad()
def synapse_test_single_volume(image, label, net, classes: int, patch_size: Sequence[int] = (256, 256), test_save_path=None,
                               case=None, z_spacing: float = 1.0, batch_slices: int = 32, device=None):
    """test_single_volume of utils_synapse.py:49-97: every slice of the volume [1, D, H, W] is resized to `patch_size` (cubic
    zoom), normalised with Normalize([0.5], [0.5]) (`ToTensor` leaves float arrays unscaled), classified, and the prediction
    resized back with order 0; returns [(dice, hd95)] for classes 1..classes-1.  A 2-D input takes the reference's second branch
    (one forward, no normalisation).  The slices go through the network in batches (with the reference's batch-1 semantics),
    the metrics run on the device.  `test_save_path` (NIfTI files through SimpleITK, which is absent here) writes .npy volumes
    with the spacing (1, 1, z_spacing) in a side-car text file instead."""
    from scipy.ndimage import zoom
    image = image.squeeze(0).cpu().numpy() if torch.is_tensor(image) else np.asarray(image)
    label = label.squeeze(0).cpu().numpy() if torch.is_tensor(label) else np.asarray(label)
    device = device or next(net.parameters()).device
    net.eval()
    if image.ndim == 3:
        D, x, y = image.shape
        resize = (x != patch_size[0] or y != patch_size[1])
        prediction = np.zeros_like(label)
        for s in range(0, D, batch_slices):
            sl = image[s:s + batch_slices]
            if resize:
                sl = np.stack([zoom(a, (patch_size[0] / x, patch_size[1] / y), order=3) for a in sl])
            inp = torch.from_numpy(np.ascontiguousarray(sl)).unsqueeze(1).float()
            inp = ((inp - 0.5) / 0.5).to(device)
            with ops.batch1_semantics():
                logits = net(inp)
            pred, _ = predict_counts(logits)
            out = pred.reshape(-1, patch_size[0], patch_size[1]).cpu().numpy()
            for i in range(out.shape[0]):
                prediction[s + i] = zoom(out[i], (x / patch_size[0], y / patch_size[1]), order=0) if resize else out[i]
    else:
        inp = torch.from_numpy(np.ascontiguousarray(image)).unsqueeze(0).unsqueeze(0).float().to(device)
        pred, _ = predict_counts(net(inp))
        prediction = pred.squeeze(0).cpu().numpy().astype(label.dtype)
    pred_d = torch.from_numpy(np.ascontiguousarray(prediction)).to(device)
    lab_d = torch.from_numpy(np.ascontiguousarray(label)).to(device)
    metric_list = [synapse_metric_percase(pred_d == c, lab_d == c) for c in range(1, classes)]
    if test_save_path is not None:
        import os
        os.makedirs(test_save_path, exist_ok=True)
        for tag, vol in (("pred", prediction), ("img", image), ("gt", label)):
            np.save(os.path.join(test_save_path, f"{case}_{tag}.npy"), np.asarray(vol, dtype=np.float32))
        with open(os.path.join(test_save_path, f"{case}_spacing.txt"), "w") as f:
            f.write(f"1 1 {z_spacing}\n")
    return metric_list


def _binary_dc(result: torch.Tensor, reference: torch.Tensor) -> float:
    """medpy.metric.binary.dc as the skin scripts call it — on arrays of DIFFERENT rank (argmax output [B, H, W] or [H, W]
    against the label batch [B, 1, H, W]): the intersection is counted on the numpy-broadcast of the two boolean arrays, the
    sizes on each array alone; 0.0 when both are empty"""
    a, b = result != 0, reference != 0
    sa, sb = int(a.sum()), int(b.sum())
    return 2.0 * int((a & b).sum()) / float(sa + sb) if (sa + sb) else 0.0


@torch.no_grad()
def skin_val(net, vl_loader, device=None) -> float:
    """val of utils_skin.py:97-113: mean over the loader's batches of dc(argmax(softmax(logits), 1).squeeze(0), label batch)"""
    device = device or next(net.parameters()).device
    net.eval()
    dc_sum, n = 0.0, 0
    for batch in vl_loader:
        img, lab = batch["image"].to(device), batch["label"].to(device)
        pred, _ = predict_counts(net(img.float()))
        dc_sum += _binary_dc(pred.squeeze(0), lab)
        n += 1
    return dc_sum / max(n, 1)


@torch.no_grad()
def skin_test(net, te_loader, device=None) -> Tuple[float, float, float]:
    """test of utils_skin.py:131-170 (batch size 1, as the reference's test loader): (mean dice, pixel accuracy, mean IoU) with
    the ground truth read as `label[0, 0]` and calc_iou(pd > 0.5, gt > 0.5) = |A & B| / |A or B| (0.0 for an empty union,
    utils_skin.py:13-27)"""
    device = device or next(net.parameters()).device
    net.eval()
    dc_sum, correct, total, ious, n = 0.0, 0, 0, [], 0
    for batch in te_loader:
        img, lab = batch["image"].float().to(device), batch["label"].float().to(device)
        pred, _ = predict_counts(net(img))
        pd = pred.squeeze(0)
        gt = lab[0, 0]
        correct += int((pd == gt).sum())
        total += gt.numel()
        a, b = pd > 0.5, gt > 0.5
        union = int((a | b).sum())
        ious.append(int((a & b).sum()) / union if union else 0.0)
        dc_sum += _binary_dc(pd, lab)
        n += 1
    return dc_sum / max(n, 1), correct / max(total, 1), float(np.mean(ious)) if ious else 0.0
