"""On-device evaluation of the hot path: the validate() / test() forward (src/train.py:137-291, src/test.py:124-318)
with every metric the reference prints derived from ONE fp64 sums kernel and one device->host copy, instead of
the reference's seven `.item()` syncs per batch (src/train.py:525-541).

As in the reference, modules stay in train mode (BatchNorm uses batch statistics and updates its running stats:
SURVEY D5) and the whole design is evaluated as one batch over all of its paths (src/train.py:188, src/test.py:176).
`endpoint_slack_mae` = mean |(required - y_hat) - (required - arrival)| = mean |y_hat - arrival| is the accuracy
metric BASELINE.json names.
"""
import numpy as np
import torch
from . import lib, ops


def eval_sums(pred, arrival, required, label):
    for t, nm in ((pred, 'pred'), (arrival, 'arrival'), (required, 'required'), (label, 'label')):
        ops._chk(t, nm)
        if t.dim() != 1 or not t.is_contiguous() or t.numel() != pred.numel():
            raise ValueError(f'eval_sums: {nm} must be a contiguous 1-D tensor of the prediction length')
    out = torch.empty(10, dtype=torch.float64, device=pred.device)
    dev, st = lib.stream_args(pred)
    lib.call('mmft_eval_sums', pred, arrival, required, label, pred.numel(), out, dev, st)
    return out


def metrics_from_sums(s):
    """s: the 10 sums of mmft_eval_sums (host floats). Same formulas as src/train.py:230-278 / torchmetrics R2Score."""
    n, sy, syy, sse, sae, sape, tp, fp, tn, fn = [float(v) for v in s]
    ss_tot = syy - sy * sy / n
    recall = tp / (tp + fn) if tp else 0.0
    precision = tp / (tp + fp) if tp else 0.0
    f1 = 2 * recall * precision / (recall + precision) if (precision or recall) else 0.0
    return dict(n=int(n), loss=sse / n, r2=1.0 - sse / ss_tot if ss_tot > 0 else float('nan'),
                endpoint_slack_mae=sae / n, mape=sape / n, acc=(tp + tn) / n, recall=recall, precision=precision,
                f1=f1, tp=int(tp), fp=int(fp), tn=int(tn), fn=int(fn))


@torch.no_grad()
def validate(train_step, path_ids_per_design=None):
    """Forward over all (or the given) paths of the designs held by `train_step` (a mmft.train.TrainStep) and
    return the metric dict.  One device->host copy."""
    b = train_step.batch
    if path_ids_per_design is None:
        path_ids_per_design = [np.arange(d.num_paths) for d in b.designs]
    hats, ends_d, _ = train_step.forward(path_ids_per_design)
    idx = ends_d.long()
    arrival = b.arrival[idx].squeeze(-1).contiguous()
    required = b.required[idx].squeeze(-1).contiguous()
    label = b.graph.ndata['label'][idx].squeeze(-1).to(torch.float32).contiguous()
    return metrics_from_sums(eval_sums(hats.contiguous(), arrival, required, label).cpu().tolist())
