"""DECODA text format reader -- counterpart of working_example.py:dataPrepDecodaQuaternion (:19-66).

One document per line: 250 space-separated `r,i,j,k` tokens, a TAB, 8 space-separated `l,l,l,l`
label tokens of which the first value is kept.  Returns float64 arrays like the reference:
x (N, 250, 4) for isquat (all components) or (N, 250, 3) (components 1..3), y (N, 8) one-hot.
"""
import numpy as np


def dataPrepDecodaQuaternion(filename, isquat=True):
    nb_topics, nb_classes = 250, 8
    with open(filename, 'r') as f:
        raw = f.readlines()
    x = np.zeros((len(raw), nb_topics, 4 if isquat else 3))
    y = np.zeros((len(raw), nb_classes))
    for d, doc in enumerate(raw):
        feats, labels = doc.split('\t')[0].split(' '), doc.split('\t')[1].split(' ')
        for e, element in enumerate(feats):
            comp = element.split(',')
            x[d, e] = [float(c) for c in (comp[:4] if isquat else comp[1:4])]
        for l, label in enumerate(labels):
            y[d, l] = float(label.split(',')[0])
    return x, y
