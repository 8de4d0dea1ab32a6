"""ImageNet label maps that ship with the Inception-v3 tarball (``retrain*/inception_model/``).

The reference carries ``imagenet_2012_challenge_label_map_proto.pbtxt`` (``entry { target_class: N
target_class_string: "n0…" }``) and ``imagenet_synset_to_human_label_map.txt`` (``nXXXXXXXX\\thuman label``) but none
of its scripts reads them — they belong to the 1008-way ``softmax`` of the frozen graph.  With the classifier
weights imported by ``utils.graphdef.inception_classifier_from_graphdef`` this lookup turns a logit index into a
human-readable name, which is what TensorFlow's own ``classify_image.py`` does with the same two files.
"""
from __future__ import annotations

import re
from typing import Dict, List, Optional, Tuple


def parse_label_map_pbtxt(path: str) -> Dict[int, str]:
    """target_class (1..1000 node id of the graph's softmax) -> synset uid."""
    out: Dict[int, str] = {}
    cls: Optional[int] = None
    with open(path, "r", encoding="utf-8") as f:
        for line in f:
            line = line.strip()
            m = re.match(r"target_class:\s*(\d+)", line)
            if m:
                cls = int(m.group(1))
                continue
            m = re.match(r'target_class_string:\s*"([^"]+)"', line)
            if m and cls is not None:
                out[cls] = m.group(1)
                cls = None
    return out


def parse_synset_to_human(path: str) -> Dict[str, str]:
    out: Dict[str, str] = {}
    with open(path, "r", encoding="utf-8") as f:
        for line in f:
            line = line.rstrip("\n")
            if not line:
                continue
            uid, _, human = line.partition("\t")
            out[uid] = human
    return out


class NodeLookup:
    """node id -> human label (unknown ids map to '')."""

    def __init__(self, label_map_pbtxt: str, synset_to_human_txt: str):
        uid_of = parse_label_map_pbtxt(label_map_pbtxt)
        human = parse_synset_to_human(synset_to_human_txt)
        missing = [u for u in uid_of.values() if u not in human]
        if missing:
            raise ValueError("synsets without a human label: %s" % ", ".join(missing[:5]))
        self.names: Dict[int, str] = {node: human[uid] for node, uid in uid_of.items()}

    def id_to_string(self, node_id: int) -> str:
        return self.names.get(int(node_id), "")

    def top_k(self, probabilities, k: int = 5) -> List[Tuple[int, str, float]]:
        """[(node id, label, score)] of the k largest entries of a 1-d probability vector (list / numpy / torch)."""
        scores = [float(v) for v in probabilities]
        order = sorted(range(len(scores)), key=lambda i: -scores[i])[:k]
        return [(i, self.id_to_string(i), scores[i]) for i in order]
