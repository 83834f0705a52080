import os

from barbell_amd import _abi, kits

EX = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "examples")


def config_groups(name):
    """Query groups of the BASELINE.json configs."""
    if name == "nbd96":  # configs[1]/[2]: SQK-NBD114-96, --flank-max-errors 3
        return kits.groups_from_kit("SQK-NBD114-96", flank_max_errors=3)
    if name == "rbk24":  # configs[0]: SQK-RBK114-24, automatic flank cutoff
        return kits.groups_from_kit("SQK-RBK114-24")
    if name == "dual":  # configs[3]: custom dual-end, --flank-max-errors 5
        return [kits.group_from_fasta(os.path.join(EX, "native_left.fasta"), _abi.BB_FTAG, 5),
                kits.group_from_fasta(os.path.join(EX, "native_right.fasta"), _abi.BB_RTAG, 5)]
    if name == "rbk96x":  # configs[4] made meaningful: SQK-RBK114-96 --use-extended (2 groups)
        return kits.groups_from_kit("SQK-RBK114-96", use_extended=True)
    if name == "nbd96x":  # configs[4] literally: --use-extended is a no-op for SQK-NBD114-96
        return kits.groups_from_kit("SQK-NBD114-96", use_extended=True, flank_max_errors=3)
    raise KeyError(name)
