"""List-of-lists restatement of FeatureExtractor::computeKeyPointsOctTree (reference src/feature_detection.cpp:833-1122;
ExtractorNode::DivideNode, include/hso/feature_detection.h:217-272), written the reference's way: every node owns a list of
keys, children are pushed to the front of a node list.  The one rule the reference leaves to the heap (the order of
equal-sized nodes in std::sort over (size, pointer)) is fixed to creation order.  Test infrastructure only (like everything
under oracle/): tests/test_octree.py checks the product's index-range implementation against it on the CPU, and
tests/test_chain_gpu.py replays every oct-tree call of a driver run through it.  Parity unpinned by the reference (it holds no
vectors for this function)."""
import math

import numpy as np

OCCUR = 3     # hso_amd.capi.KP_OCCUR


class Node:
    serial = 0

    def __init__(self, x0, y0, x1, y1):
        self.x0, self.y0, self.x1, self.y1 = x0, y0, x1, y1
        self.keys, self.no_more = [], False
        Node.serial += 1
        self.serial = Node.serial

    def divide(self):
        half_x = math.ceil(np.float32(self.x1 - self.x0) / 2)
        half_y = math.ceil(np.float32(self.y1 - self.y0) / 2)
        mx, my = self.x0 + half_x, self.y0 + half_y
        n1, n2 = Node(self.x0, self.y0, mx, my), Node(mx, self.y0, self.x1, my)
        n3, n4 = Node(self.x0, my, mx, self.y1), Node(mx, my, self.x1, self.y1)
        for k in self.keys:
            if k["x"] < mx:
                (n1 if k["y"] < my else n3).keys.append(k)
            else:
                (n2 if k["y"] < my else n4).keys.append(k)
        for c in (n1, n2, n3, n4):
            c.no_more = len(c.keys) == 1
        return n1, n2, n3, n4


def octree_py(keys, width, height, n_features):
    n_ini = int(round(float(np.float32(width) / np.float32(height))))
    hx = np.float32(width) / np.float32(n_ini)
    ini = [Node(int(hx * np.float32(i)), 0, int(hx * np.float32(i + 1)), height) for i in range(n_ini)]
    for k in keys:
        ini[int(np.float32(int(k["x"])) / hx)].keys.append(k)
    nodes = []
    for nd in ini:
        if len(nd.keys) == 1:
            nd.no_more = True
        if nd.keys:
            nodes.append(nd)
    finish = False
    while not finish:
        prev = len(nodes)
        front, keep, expandable = [], [], []
        for nd in nodes:                                   # one sweep over the nodes that exist now
            if nd.no_more:
                keep.append(nd)
                continue
            for c in nd.divide():
                if c.keys:
                    front.insert(0, c)                     # push_front
                    if len(c.keys) > 1:
                        expandable.append(c)
        nodes = front + keep
        if len(nodes) >= n_features or len(nodes) == prev:
            finish = True
        elif len(nodes) + 3 * len(expandable) > n_features:
            while not finish:
                prev = len(nodes)
                todo = sorted(expandable, key=lambda c: (len(c.keys), c.serial))
                expandable = []
                for nd in reversed(todo):
                    for c in nd.divide():
                        if c.keys:
                            nodes.insert(0, c)
                            if len(c.keys) > 1:
                                expandable.append(c)
                    nodes.remove(nd)
                    if len(nodes) >= n_features:
                        break
                if len(nodes) >= n_features or len(nodes) == prev:
                    finish = True
    out = []
    for nd in nodes:
        best = nd.keys[0]
        if best["species"] == OCCUR:
            continue
        occur = False
        for k in nd.keys[1:]:
            if k["species"] == OCCUR:
                occur = True
                break
            if best["species"] > k["species"] or (best["species"] == k["species"] and k["response"] > best["response"]):
                best = k
        if not occur:
            out.append(best)
    return out
