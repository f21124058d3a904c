"""SMILES front end (bmp/smiles.py): hand-derived cases only -- RDKit, the reference's parser (parsers.py:209-210), is
not available, so parity with it is UNPINNED; these pin the grammar and the bond-channel conventions."""
import numpy as np
import pytest

from bmp.smiles import AROMATIC, DOUBLE, SINGLE, TRIPLE, SmilesError, parse_pair_csv, parse_smiles, smiles_to_molecule


def bonds_of(s):
    atoms, bonds = parse_smiles(s)
    return atoms.tolist(), sorted((min(a, b), max(a, b), t) for a, b, t in bonds.tolist())


def test_chains_branches_and_bond_orders():
    assert bonds_of("CCO") == ([6, 6, 8], [(0, 1, SINGLE), (1, 2, SINGLE)])
    assert bonds_of("C#N") == ([6, 7], [(0, 1, TRIPLE)])
    assert bonds_of("CC(C)(C)C(=O)O") == ([6, 6, 6, 6, 6, 8, 8],
                                           [(0, 1, SINGLE), (1, 2, SINGLE), (1, 3, SINGLE), (1, 4, SINGLE), (4, 5, DOUBLE), (4, 6, SINGLE)])
    assert bonds_of("F/C=C\\Cl") == ([9, 6, 6, 17], [(0, 1, SINGLE), (1, 2, DOUBLE), (2, 3, SINGLE)])
    assert bonds_of("BrCCl")[0] == [35, 6, 17]


def test_rings_and_aromatic_bonds():
    atoms, b = bonds_of("c1ccccc1")
    assert atoms == [6] * 6 and all(t == AROMATIC for _, _, t in b) and len(b) == 6
    # Kekule form is aromatised as RDKit's sanitisation would
    _, b = bonds_of("C1=CC=CC=C1")
    assert [t for _, _, t in b] == [AROMATIC] * 6
    # biphenyl: the unmarked bond between the rings is a bridge -> single
    _, b = bonds_of("c1ccccc1c1ccccc1")
    assert sum(t == SINGLE for _, _, t in b) == 1 and sum(t == AROMATIC for _, _, t in b) == 12
    assert (5, 6, SINGLE) in b
    # fused rings (naphthalene): the shared bond is in a ring -> aromatic
    _, b = bonds_of("c1ccc2ccccc2c1")
    assert len(b) == 11 and all(t == AROMATIC for _, _, t in b)
    # pyridine-like nitrogen, furan oxygen, bracket aromatic [nH]
    assert bonds_of("c1ccncc1")[0] == [6, 6, 6, 7, 6, 6]
    assert bonds_of("c1cc[nH]c1")[0] == [6, 6, 6, 7, 6]
    # ring closure carrying the bond symbol, and two-digit closures
    _, b = bonds_of("C=1CCCCC=1")
    assert (0, 5, DOUBLE) in b
    _, b = bonds_of("C%12CC%12")
    assert (0, 2, SINGLE) in b and len(b) == 3


def test_brackets_charges_isotopes_dots_and_hydrogens():
    assert bonds_of("[Na+].[Cl-]") == ([11, 17], [])
    assert bonds_of("[13CH4]") == ([6], [])
    assert bonds_of("C[N+](C)(C)C")[0] == [6, 7, 6, 6, 6]
    assert bonds_of("[Se]=O")[0] == [34, 8]
    assert bonds_of("[H]C([H])=O") == ([6, 8], [(0, 1, DOUBLE)])        # explicit hydrogens on a heavy atom are dropped
    assert bonds_of("[H][H]") == ([1, 1], [(0, 1, SINGLE)])             # molecular hydrogen stays
    assert bonds_of("C[C@@H](N)C(=O)O")[0] == [6, 6, 7, 6, 8, 8]


def test_dense_arrays_follow_the_preprocessor_contract():
    m = smiles_to_molecule("CC(=O)Oc1ccccc1C(=O)O")                      # aspirin, aromatic form
    adj = m.dense_adj()
    assert m.atoms.tolist() == [6, 6, 8, 8, 6, 6, 6, 6, 6, 6, 6, 8, 8]
    assert adj.shape == (4, 13, 13) and np.array_equal(adj, adj.transpose(0, 2, 1)) and adj.diagonal(axis1=1, axis2=2).sum() == 0
    assert adj[AROMATIC].sum() == 12 and adj[DOUBLE].sum() == 4 and adj[TRIPLE].sum() == 0 and adj[SINGLE].sum() == 10
    with pytest.raises(SmilesError):
        smiles_to_molecule("CCCC", max_atoms=3)


@pytest.mark.parametrize("bad", ["C1CC", "C(C", "CC)", "[C", "C$C", "*C", "", "Xx", "C11"])
def test_malformed_input_raises(bad):
    with pytest.raises(SmilesError):
        parse_smiles(bad)


def test_pair_csv_builds_a_store_and_index_pairs(tmp_path):
    p = tmp_path / "pairs.csv"
    p.write_text("smiles_1,smiles_2,label\nCCO,c1ccccc1,1\nc1ccccc1,CC(=O)O,0\nCCO,C1CC,1\nCCO,CCO,0\n")
    r = parse_pair_csv(str(p), labels=["label"])
    assert r["n_failed"] == 1 and r["smiles"] == ["CCO", "c1ccccc1", "CC(=O)O"]
    assert r["idx1"].tolist() == [0, 1, 0] and r["idx2"].tolist() == [1, 2, 0] and r["labels"].ravel().tolist() == [1, 0, 0]
    assert [m.n for m in r["store"]] == [3, 6, 4]
    # and the store packs like the synthetic one
    from bmp import packed
    pb = packed.pack_from_store(packed.MolStore(r["store"]), [r["idx1"], r["idx2"]])
    assert pb.n_mols == 6 and pb.n_real_atoms == 3 + 6 + 3 + 6 + 4 + 3


def _types(smiles):
    return sorted(t for _a, _b, t in parse_smiles(smiles)[1].tolist())


def test_aromaticity_perception_of_kekule_input():
    """Electron counting per ring (bmp/smiles.py:perceive_aromaticity); hand-derived expectations = what RDKit's default
    model gives for these textbook cases."""
    same = lambda kek, aro: _types(kek) == _types(aro) and parse_smiles(kek)[0].tolist() == parse_smiles(aro)[0].tolist()
    assert same("C1=CC=CC=C1", "c1ccccc1")                          # benzene
    assert same("C1=CC=NC=C1", "c1ccncc1")                          # pyridine
    assert same("C1=CNC=C1", "c1c[nH]cc1")                          # pyrrole: N-H brings two electrons
    assert same("C1=COC=C1", "c1cocc1") and same("C1=CSC=C1", "c1cscc1")      # furan, thiophene
    assert same("C1=CN=CN1", "c1cnc[nH]1")                          # imidazole
    assert same("C1=CC=C2C=CC=CC2=C1", "c1ccc2ccccc2c1")            # naphthalene, fused bond single in this Kekule form
    assert same("C1=CC2=CC=CC=C2C=C1", "c1ccc2ccccc2c1")            # ... and the other form
    assert same("C1=CC=C2NC=CC2=C1", "c1ccc2[nH]ccc2c1")            # indole
    assert same("CC(=O)OC1=CC=CC=C1C(=O)O", "CC(=O)Oc1ccccc1C(=O)O")      # aspirin
    assert same("O=C1C=CC=CN1", "O=c1cccc[nH]1")                    # 2-pyridone: the exocyclic C=O carbon brings none
    assert same("CN1C=NC2=C1C(=O)N(C)C(=O)N2C", "Cn1cnc2c1c(=O)n(C)c(=O)n2C")     # caffeine
    # not aromatic: 4n electrons, sp3 ring atoms, isolated double bonds
    assert _types("C1=CCC=C1") == [SINGLE] * 3 + [DOUBLE] * 2       # cyclopentadiene (CH2 in the ring)
    assert _types("C1=CCCCC1") == [SINGLE] * 5 + [DOUBLE]           # cyclohexene
    assert _types("O=C1C=CC(=O)C=C1") == [SINGLE] * 4 + [DOUBLE] * 4      # p-benzoquinone: 4 electrons
    assert _types("C1=CC=CC=CC=C1") == [SINGLE] * 4 + [DOUBLE] * 4  # cyclooctatetraene (8 electrons, and ring size 8)
    assert _types("C1CCCCC1") == [SINGLE] * 6
    # aromatic input is left as written; mixing notations in one ring leaves the ring alone
    assert _types("c1ccccc1") == [AROMATIC] * 6


def test_exocyclic_carbon_carbon_double_bonds_do_not_aromatise_a_ring():
    """A double bond counts towards a ring's pi electrons only if the BOND lies in a ring: fulvene's exocyclic C=C and the
    ylidene link of fulvalene leave their five-rings Kekule (hand-derived; RDKit reports no aromatic atoms for either)."""
    _, b = bonds_of("C=C1C=CC=C1")                                      # fulvene
    assert sum(t == AROMATIC for _, _, t in b) == 0 and sum(t == DOUBLE for _, _, t in b) == 3
    _, b = bonds_of("C1=CC=CC1=C1C=CC=C1")                              # fulvalene: inter-ring C=C between two five-rings
    assert sum(t == AROMATIC for _, _, t in b) == 0 and sum(t == DOUBLE for _, _, t in b) == 5
    # ... while an exocyclic C=O still leaves its ring free to be aromatic (2-pyridone) and benzene rings next to an
    # exocyclic C=C keep theirs (styrene)
    _, b = bonds_of("O=C1C=CC=CN1")
    assert sum(t == AROMATIC for _, _, t in b) == 6
    _, b = bonds_of("C=CC1=CC=CC=C1")
    assert sum(t == AROMATIC for _, _, t in b) == 6 and sum(t == DOUBLE for _, _, t in b) == 1
