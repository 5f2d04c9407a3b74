"""The shipped training criteria and their combination (training row of SURVEY 8(f4)).

ref: pointcept/models/losses/builder.py:14-52 (Criteria: "EW" sums the terms, "GLS" with task_num = 2 takes
sqrt(MSE * (CE + Lovasz))), losses/misc.py:24-93 (MSELoss on the noise branch over the labelled points),
misc.py:95-132 (CrossEntropyLoss), losses/lovasz.py:118-165, 210-265 (multi-class Lovasz-Softmax over the classes
present).  Every config of the reference ships exactly these three (configs/*/CDSegNet.py:117-123).  Plain torch ops on
the prediction tensors: the loss is a few reductions over (N, classes) - the kernels of the path are below it, in
cdsegnet_amd.train_graph.  Other criterion types are rejected loudly.
"""
import torch
import torch.nn.functional as F


def _valid(point, ignore_index, key="n_target"):
    return point[key] != ignore_index


class MSELoss:
    def __init__(self, pred="c_pred", target="c_target", segment_target="n_target", batch_sample_point=8192, reduction="none",
                 loss_weight=1.0, ignore_index=None, **unused):
        if reduction != "none":
            raise NotImplementedError(f"MSELoss(reduction={reduction!r})")
        self.pred, self.target, self.segment_target = pred, target, segment_target
        self.bsp, self.loss_weight, self.ignore_index = batch_sample_point, loss_weight, ignore_index

    def __call__(self, point):
        if self.pred not in point or self.target not in point:
            return 0.0
        pred, target = point[self.pred], point[self.target]
        if self.bsp > 0:  # per-scene subsampling (misc.py:59-75); the shipped configs pass -1
            ps, ts, start = [], [], 0
            for end in point["offset"].tolist():
                p, t = pred[start:end], target[start:end]
                if self.bsp < end - start:
                    ch = torch.randint(low=0, high=end - start, size=(self.bsp,)).to(p.device)
                    p, t = p[ch], t[ch]
                ps.append(p)
                ts.append(t)
                start = end
            pred, target = torch.cat(ps, 0), torch.cat(ts, 0)
        if self.ignore_index:  # (truthiness as in the reference: ignore_index = 0 / None switch the filter off)
            valid = _valid(point, self.ignore_index, self.segment_target)
            pred, target = pred[valid], target[valid]
        return ((pred - target) ** 2).mean() * self.loss_weight


class CrossEntropyLoss:
    def __init__(self, pred="n_pred", target="n_target", weight=None, reduction="mean", label_smoothing=0.0, loss_weight=1.0,
                 ignore_index=-1, **unused):
        self.pred, self.target, self.weight = pred, target, weight
        self.reduction, self.label_smoothing, self.loss_weight, self.ignore_index = reduction, label_smoothing, loss_weight, ignore_index

    def __call__(self, point):
        if self.pred not in point or self.target not in point:
            return 0.0
        pred, target = point[self.pred], point[self.target]
        if self.ignore_index:
            valid = target != self.ignore_index
            pred, target = pred[valid], target[valid]
        w = None if self.weight is None else torch.as_tensor(self.weight, dtype=pred.dtype, device=pred.device)
        return F.cross_entropy(pred, target, weight=w, reduction=self.reduction, label_smoothing=self.label_smoothing) * self.loss_weight


def lovasz_softmax(prob, labels, ignore=None):
    """Multi-class Lovasz-Softmax, classes present, whole batch (lovasz.py:118-165 with per_image=False, class_seen=None)."""
    if ignore is not None:
        valid = labels != ignore
        prob, labels = prob[valid], labels[valid]
    if prob.numel() == 0:
        return prob.sum() * 0.0
    terms = []
    for c in torch.unique(labels).tolist():
        fg = (labels == c).to(prob.dtype)
        err = (fg - prob[:, c]).abs()
        err_s, order = torch.sort(err, 0, descending=True)
        fg_s = fg[order]
        total = fg_s.sum()
        jac = 1.0 - (total - fg_s.cumsum(0)) / (total + (1.0 - fg_s).cumsum(0))
        if len(jac) > 1:
            jac = torch.cat([jac[:1], jac[1:] - jac[:-1]])
        terms.append(torch.dot(err_s, jac))
    return torch.stack(terms).mean()


class LovaszLoss:
    def __init__(self, mode, pred="n_pred", target="n_target", class_seen=None, per_image=False, ignore_index=None,
                 loss_weight=1.0, **unused):
        if mode != "multiclass" or per_image or class_seen is not None:
            raise NotImplementedError(f"LovaszLoss(mode={mode!r}, per_image={per_image}, class_seen={class_seen}): the shipped "
                                      "configs use mode='multiclass' over the whole batch")
        self.pred, self.target, self.ignore_index, self.loss_weight = pred, target, ignore_index, loss_weight

    def __call__(self, point):
        if self.pred not in point or self.target not in point:
            return 0.0
        return lovasz_softmax(point[self.pred].softmax(dim=1), point[self.target], self.ignore_index) * self.loss_weight


_TYPES = {"MSELoss": MSELoss, "CrossEntropyLoss": CrossEntropyLoss, "LovaszLoss": LovaszLoss}


class Criteria:
    """ref: losses/builder.py:14-52."""

    def __init__(self, cfg=None, loss_type="EW", task_num=2):
        self.criteria = []
        for c in (cfg or []):
            c = dict(c)
            t = c.pop("type")
            if t not in _TYPES:
                raise NotImplementedError(f"criterion {t!r}: the shipped CDSegNet configs use MSELoss, CrossEntropyLoss, LovaszLoss")
            self.criteria.append(_TYPES[t](**c))
        self.loss_type, self.task_num = loss_type, task_num

    def __call__(self, point):
        if not self.criteria:
            return point
        mode = point["loss_mode"]
        if mode == "eval" or self.loss_type == "EW":
            loss = 0.0
            for c in self.criteria:
                loss = loss + c(point)
            return loss
        if mode == "train" and self.loss_type == "GLS":
            parts = [c(point) for c in self.criteria]
            if self.task_num == 1:
                loss = parts[0] + parts[1]
            elif self.task_num == 2 and self.task_num != len(parts):
                loss = parts[0] * (parts[1] + parts[2])  # MSE x (cross entropy + Lovasz)
            else:
                raise NotImplementedError(f"GLS with task_num={self.task_num} and {len(parts)} criteria")
            return torch.pow(loss, 1.0 / self.task_num)
        return 0.0  # (builder.py:27: any other combination leaves the sum at its initial value)


def build_criteria(cfg, loss_type="EW", task_num=2):
    return Criteria(cfg, loss_type=loss_type, task_num=task_num)
