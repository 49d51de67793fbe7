"""
Predicate statistics of a training set, the table FrequencyBias is initialised from (reference
lib/get_dataset_counts.py:12-67: `get_counts`, `box_filter`).

    fg_matrix[o1, o2, p] = number of ground-truth relations (subject class o1, object class o2, predicate p)
    bg_matrix[o1, o2]    = number of ordered box pairs of those classes that are candidate relations: every pair of
                           distinct OVERLAPPING boxes of an image (must_overlap; all ordered pairs when nothing in the
                           image overlaps), or simply all ordered pairs of distinct boxes
Works on anything with the dataset attributes `gt_classes`, `relationships`, `gt_boxes`, `num_classes`,
`num_predicates` (dataloaders.visual_genome.VG, dataloaders.synthetic.SyntheticVG).  The counts are accumulated with
np.add.at (one vectorised scatter per image) instead of the reference's Python loop over relations.
"""
import numpy as np

from lib.fpn.box_intersections_cpu.bbox import bbox_overlaps


def box_filter(boxes, must_overlap=False):
    """ordered index pairs (i, j), i != j, that count as possible relations (reference :49-67)"""
    n = boxes.shape[0]
    off_diag = ~np.eye(n, dtype=bool)
    if must_overlap:
        b = boxes.astype(np.float64)
        pairs = np.column_stack(np.where((bbox_overlaps(b, b) > 0) & off_diag))
        if pairs.size:
            return pairs
    return np.column_stack(np.where(off_diag))


def get_counts(train_data, must_overlap=True):
    fg_matrix = np.zeros((train_data.num_classes, train_data.num_classes, train_data.num_predicates), dtype=np.int64)
    bg_matrix = np.zeros((train_data.num_classes, train_data.num_classes), dtype=np.int64)
    for ex_ind in range(len(train_data)):
        gt_classes = np.asarray(train_data.gt_classes[ex_ind])
        gt_relations = np.asarray(train_data.relationships[ex_ind])
        if gt_relations.size:
            o1o2 = gt_classes[gt_relations[:, :2]]
            np.add.at(fg_matrix, (o1o2[:, 0], o1o2[:, 1], gt_relations[:, 2]), 1)
        cand = box_filter(np.asarray(train_data.gt_boxes[ex_ind]), must_overlap=must_overlap)
        if cand.size:
            o1o2 = gt_classes[cand.astype(np.int64)]
            np.add.at(bg_matrix, (o1o2[:, 0], o1o2[:, 1]), 1)
    return fg_matrix, bg_matrix
