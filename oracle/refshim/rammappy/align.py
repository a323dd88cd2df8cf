from types import SimpleNamespace


class Aligner:
    def __init__(self, index=None, preset=None, do_cigar=True, do_cs=False, do_md=False):
        self.index = index
        self.options = SimpleNamespace(filtering=SimpleNamespace(best_n=5, pri_ratio=0.8))

    def map_batch(self, queries):
        import rammappy

        return [iter(rammappy.PENDING_HITS.get(int(name), ())) for name, _seq in queries]
