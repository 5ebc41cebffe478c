"""Field-level accuracy of a batch, mirror of ``cal_accuracy`` in /root/reference/utils.py:14-92
(SURVEY.md 8f2).  Host-side numpy on the arrays the step returns; same return tuple
``(prevalence, accuracy_strict, accuracy_soft, res)`` with ``res`` the utf-8 encoded report of
sample 0.  Pinned by tests/golden/accuracy_cases.* (outputs of the reference's own function).

Per sample and per class c >= 1, over the cells that hold a word (grid id != 0):
  predicted cells = softmax[:, c] > 0.5 (``c_threshold``, :12,43)
  strict  +1 when the set of predicted boxes (bbox_mapids of the predicted cells) equals the
          class's ground-truth boxes (label_mapids[c]); soft +1 also when it contains them (:52-56)
  prevalence += |predicted cells & labelled cells| / |labelled cells|, 1 when the class has no
          labelled cell (:57-63)
all divided by batch * (num_classes - 1) (:21,89-91).
"""
import numpy as np

c_threshold = 0.5


def cal_accuracy(data_loader, grid_table, gt_classes, model_output_val, label_mapids, bbox_mapids):
    grid_table = np.asarray(grid_table)
    gt_classes = np.asarray(gt_classes)
    model_output_val = np.asarray(model_output_val)
    num_classes = data_loader.num_classes
    words, classes = data_loader.index_to_word, data_loader.classes
    batch = grid_table.shape[0]
    num_all = batch * (model_output_val.shape[-1] - 1)
    correct = strict = soft = 0
    lines = []
    for b in range(batch):
        ids = grid_table[b, :, :, 0].reshape(-1)
        cells = np.flatnonzero(ids)                      # flat cell index = row * cols + col = the bbox key
        ids_sel = ids[cells]
        labels_sel = gt_classes[b].reshape(-1)[cells]
        probs_sel = model_output_val[b].reshape(-1, num_classes)[cells]
        bbox_mapid = bbox_mapids[b]
        for c in range(1, num_classes):
            lab_idx = np.flatnonzero(labels_sel == c)
            out_idx = np.flatnonzero(probs_sel[:, c] > c_threshold)
            want = set(label_mapids[b][c])
            got = {bbox_mapid[k] for k in cells[out_idx].tolist() if k in bbox_mapid}
            if want == got:
                strict += 1
                soft += 1
            elif want <= got:
                soft += 1
            if len(lab_idx):
                correct += np.intersect1d(lab_idx, out_idx).shape[0] / len(lab_idx)
            else:
                correct += 1
            if b == 0:
                line = '\n{}(GT/Inf):\t"'.format(classes[c])
                line += ' '.join(words[i] for i in ids_sel[lab_idx]) + '" | "'
                line += ' '.join(words[i] for i in ids_sel[out_idx]) + '"'
                if not np.array_equal(lab_idx, out_idx):
                    line += '\n \t FALSES =>>'
                    for i in np.setdiff1d(out_idx, lab_idx):
                        line += ' "%s"/%s, ' % (words[ids_sel[i]], classes[labels_sel[i]])
                lines.append(line)
    return correct / num_all, strict / num_all, soft / num_all, ''.join(lines).encode("utf-8")
