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


def eval_sums_by_level(pred, arrival, required, label, level, num_levels):
    """[num_levels, 10] fp64 sums, one row per topological level (mmft_eval_sums_by_level)."""
    for t, nm in ((pred, 'pred'), (arrival, 'arrival'), (required, 'required'), (label, 'label')):
        ops._chk(t, nm)
        if t.dim() != 1 or not t.is_contiguous() or t.numel() != pred.numel():
            raise ValueError(f'eval_sums_by_level: {nm} must be a contiguous 1-D tensor of the prediction length')
    ops._idx(level, 'level', pred.numel())
    out = torch.empty((num_levels, 10), dtype=torch.float64, device=pred.device)
    dev, st = lib.stream_args(pred)
    lib.call('mmft_eval_sums_by_level', pred, arrival, required, label, level, pred.numel(), int(num_levels), out, dev, st)
    return out


def level_metrics_from_sums(rows):
    """Per-level R2 / MAPE as src/test.py:211-216 prints them (levels with >= 2 predictions), from the [L, 10] sums."""
    out = []
    for l, srow in enumerate(rows):
        n = int(srow[0])
        if n < 2:
            continue
        m = metrics_from_sums(srow)
        out.append(dict(level=l, n=n, r2=m['r2'], mape=m['mape'], mae=m['endpoint_slack_mae']))
    return out


@torch.no_grad()
def validate(train_step, path_ids_per_design=None, per_level=False):
    """Forward over all (or the given) paths of the designs held by `train_step` (a mmft.train.TrainStep) and
    return the metric dict; per_level=True adds 'levels': R2 / MAPE of every topological level (src/test.py:211-216).
    One device->host copy."""
    b = train_step.batch
    if path_ids_per_design is None:
        path_ids_per_design = [np.arange(d.num_paths) for d in b.designs]
    sel = b.select(path_ids_per_design)
    hats, ends_d, _ = train_step.forward(path_ids_per_design, _sel=sel)
    idx = ends_d.long()
    arrival = b.arrival[idx].squeeze(-1).contiguous()
    required = b.required[idx].squeeze(-1).contiguous()
    label = b.graph.ndata['label'][idx].squeeze(-1).to(torch.float32).contiguous()
    hats = hats.contiguous()
    if getattr(train_step, 'task', 'reg') == 'cls':
        # classification task (src/train.py:516-518,536-549): CrossEntropy + argmax prediction, positive = class != 0
        from .fusion import cls_eval_sums
        labels = b.graph.ndata['label'][idx].squeeze(-1).contiguous()
        n, lsum, tp, fp, tn, fn = [float(v) for v in cls_eval_sums(hats, labels).cpu().tolist()]
        recall = tp / (tp + fn) if tp else 0.0
        precision = tp / (tp + fp) if tp else 0.0
        f1 = 2 * recall * precision / (recall + precision) if (precision or recall) else 0.0
        return dict(n=int(n), loss=lsum / n, r2=0.0, acc=(tp + tn) / n, recall=recall, precision=precision, f1=f1,
                    tp=int(tp), fp=int(fp), tn=int(tn), fn=int(fn))
    if not per_level:
        return metrics_from_sums(eval_sums(hats, arrival, required, label).cpu().tolist())
    both = torch.cat([eval_sums(hats, arrival, required, label).reshape(1, 10),
                      eval_sums_by_level(hats, arrival, required, label, sel[5].contiguous(), b.L)], 0).cpu().tolist()
    m = metrics_from_sums(both[0])
    m['levels'] = level_metrics_from_sums(both[1:])
    return m


def validate_designs(pmodel, cnn, designs, device, per_level=True, mode='sweep'):
    """The per-design loop of validate() / test() (src/train.py:137-291, src/test.py:124-318): every design is evaluated
    on its own as ONE batch over all of its paths (modules stay in train mode), the reference's per-case line
    (loss, r2, acc, recall, precision, F1 + per-level R2 / MAPE) is returned per design together with the averages over
    the designs that the loops print at the end (src/train.py:280-290)."""
    from .train import TrainStep
    cases = []
    for d in designs:
        ts = TrainStep(pmodel, cnn, [d], device, mode=mode, overlap=False, with_optimizer=False)
        cases.append(validate(ts, per_level=per_level))
    keys = ('loss', 'r2', 'acc', 'recall', 'precision', 'f1', 'endpoint_slack_mae', 'mape')
    overall = {k: float(np.mean([c[k] for c in cases])) for k in keys} if cases else {}
    return dict(cases=cases, overall=overall)
