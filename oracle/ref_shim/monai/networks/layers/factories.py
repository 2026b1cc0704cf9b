class Act:
    PRELU = "prelu"
    LEAKYRELU = "leakyrelu"


class Norm:
    INSTANCE = "instance"
    BATCH = "batch"
