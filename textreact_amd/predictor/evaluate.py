"""Top-k accuracies of the test step's predictions (textreact/evaluate.py).

`evaluate_reaction_condition` (evaluate.py:15-24) is plain bookkeeping; `evaluate_retrosynthesis`
(evaluate.py:43-71, template-free branch) compares canonical SMILES -- RDKit is imported lazily for the
canonicalisation and, like the reference's `canonical_smiles`, a string RDKit cannot parse is compared as
it is.  The template-based branch of the reference turns (site, template) edits back into SMILES with
template_decoder.py (RDKit reaction machinery on LocalRetro templates): host-side chemistry outside this
path, not rebuilt.
"""
CONDITION_COLS = ['catalyst1', 'solvent1', 'solvent2', 'reagent1', 'reagent2']   # textreact/dataset.py


def evaluate_reaction_condition(prediction, data_df, cutoffs=(1, 3, 5, 10, 15)):
    """{k: fraction of examples whose label is among the first k predictions}; the denominator is the
    whole data frame (evaluate.py:23)"""
    cnt = {x: 0 for x in cutoffs}
    for i, output in prediction.items():
        label = data_df.loc[i, CONDITION_COLS].tolist()
        hit_map = [pred == label for pred in output['prediction']]
        for x in cnt:
            cnt[x] += bool(any(hit_map[:x]))
    n = len(data_df)
    return {x: cnt[x] / n for x in cnt}


def canonical_smiles(smiles):
    """RDKit canonical form; an unparsable string stays as it is (the reference's try / except around CanonSmiles).  A
    MISSING RDKit is an error, as in the reference, whose module fails at import: comparing raw strings instead would
    report wrong accuracies without a word.  Pass `canonical=` to evaluate_retrosynthesis to use something else."""
    from rdkit import Chem      # ImportError propagates
    try:
        return Chem.CanonSmiles(smiles)
    except Exception:
        return smiles


def evaluate_retrosynthesis(prediction, gold_smiles, cutoffs=(1, 2, 3, 5, 10, 20), canonical=canonical_smiles):
    """prediction: {i: {'prediction': [smiles, ...]}} for i in range(len(gold_smiles)), best first"""
    n = len(gold_smiles)
    ranks = []
    for i in range(n):
        gold = canonical(gold_smiles[i])
        preds = [canonical(s) for s in prediction[i]['prediction']]
        ranks.append(preds.index(gold) if gold in preds else 100000)
    return {x: sum(r < x for r in ranks) / n for x in cutoffs}
