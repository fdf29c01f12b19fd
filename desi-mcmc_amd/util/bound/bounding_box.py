"""calc_bounding_radius / get_bounding_boxes_idx with the signatures of
CelestePy/util/bound/bounding_box.py:9-46."""
import numpy as np

from ... import field as _field


def calc_bounding_radius(weights, means, covars, error, center=np.array([0, 0])):
    """radius of the circle about `center` holding >= 1-error of every component's mass (:9-31)"""
    return _field.bounding_radius(weights, means, covars, error, center=center)


def get_bounding_boxes_idx(loc, boxes):
    """indices of the [minx, maxx, miny, maxy] rows containing loc (:41-46)"""
    boxes = np.asarray(boxes)
    inside = (loc[0] >= boxes[:, 0]) & (loc[0] <= boxes[:, 1]) & (loc[1] >= boxes[:, 2]) & (loc[1] <= boxes[:, 3])
    return np.where(inside)[0]
