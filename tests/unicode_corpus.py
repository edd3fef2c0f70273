"""Random text with diacritics, mixed case, every delimiter of ConfigurationParameters.cs:58-62, digits, odd whitespace, repeated words, empty
and one-character documents, duplicate texts — shared by the host-parity (CPU) and GPU-parity tests."""
import random

ALPHABET = list("abcdefghijklmnoprstuvzáčďéěíňóřšťúůýžäöüßÀÉÎÕÜñçøåæœABCDEFGHIJKLMNOPRSTUVZÁČĎÉĚÍŇÓŘŠŤÚŮÝŽ0123456789")
DELIMS = [" ", " ", " ", " ", "-", "/", ".", ",", ":", ";", "'", "`", "–", "—", "*", "&", "\\", "_", "(", ")", "{", "}", "[", "]", "\t",
          " ", "  ", "\n", "§", "!"]


# Scripts beyond Latin-1 / Latin Extended-A / basic Greek and Cyrillic (round 5: the case tables and the letter set come from Unicode data): Vietnamese
# (Latin Extended Additional), Latin Extended-B digraphs incl. the title-case forms, accented and final-sigma Greek, Cyrillic beyond U+045F, Armenian,
# Georgian (Asomtavruli -> Nuskhuri), full-width Latin, Cherokee, and the characters OrdinalIgnoreCase equates with another lower-case letter
# (micro sign / mu, long s / s, final sigma / sigma, the Greek symbol forms).
SCRIPTS = {
    "vietnamese": list("ảẢấẤầẦẩẨẫẪậẬắẮằẰẳẲẵẴặẶẹẸẻẺẽẼếẾềỀểỂễỄệỆỉỈịỊọỌỏỎốỐồỒổỔỗỖộỘớỚờỜởỞỡỠợỢụỤủỦứỨừỪửỬữỮựỰỳỲỵỴỷỶỹỸđĐ"),
    "latin_ext_b": list("ǄǅǆǇǈǉǊǋǌƀɃƁɓƂƃƇƈȘșȚțǍǎǏǐǑǒǓǔǕǖǞǟǺǻǼǽǾǿȀȁȦȧȲȳ"),
    # (without final sigma and the symbol forms: they share their capitals with sigma, beta, theta ... — see "aliases" below)
    "greek": list("ΆάΈέΉήΊίΌόΎύΏώΪϊΫϋΐΰαβγδεζηθικλμνξοπρστυφχψωΑΒΓΔΕΖΗΘΙΚΛΜΝΞΟΠΡΣΤΥΦΧΨΩ"),
    "cyrillic_ext": list("ѠѡѢѣѤѥѦѧѪѫҊҋҌҍҐґҒғҖҗҚқҢңҮүҰұҲҳҺһӀӏӁӂӐӑӒӓӘәӨөԀԁԐԑԚԛԜԝ"),
    "armenian": list("ԱԲԳԴԵԶԷԸԹԺԻԼԽԾԿՀՁՂՃՄՅՆՇՈՉՊՋՌՍՎՏՐՑՒՓՔՕՖաբգդեզէըթժիլխծկհձղճմյնշոչպջռսվտրցւփքօֆև"),
    "georgian": list("ႠႡႢႣႤႥႦႧႨႩႪႫႬႭႮႯႰႱႲႳႴႵⴀⴁⴂⴃⴄⴅⴆⴇⴈⴉⴊⴋⴌⴍⴎⴏⴐⴑⴒⴓⴔⴕაბგდევზთი"),
    "fullwidth": list("ＡＢＣＤＥＦＧＨＩＪＫＬＭＮＯＰＱＲＳＴＵＶＷＸＹＺａｂｃｄｅｆｇｈｉｊｋｌｍｎｏｐｑｒｓｔｕｖｗｘｙｚ０１２"),
    "cherokee": list("ᎠᎡᎢᎣᎤᎥᎦᎧᎨᎩꭰꭱꭲꭳꭴꭵꭶꭷꭸꭹᏰᏱᏲᏳᏴᏵᏸᏹᏺᏻᏼᏽ"),
    # the lower-case characters OrdinalIgnoreCase equates with ANOTHER lower-case letter (long s, micro sign, final sigma, the Greek symbol forms, ypogegrammeni,
    # long s with dot, Cyrillic Extended-C) WITHOUT their base letters: alone they behave like any other letter at every comparison site; where an alias meets
    # its base letter the reference's OrdinalIgnoreCase sites and its ToLowerInvariant sites disagree with each other (tests/test_gpu_parity.py::test_ordinal_ignore_case_aliases)
    "aliases": list("ſµςϐϑϕϖϰϱϵẛᲀᲁᲂᲃᲄᲆᲇxyzXYZ"),
}


def make_script(seed, script, ndocs=300, nqueries=50):
    """A corpus whose words are drawn from ONE script's alphabet mixed with ASCII (so that case pairs, title-case forms and alias characters meet)."""
    # (no upper-cased documents for the alias characters: their capitals are the BASE letters' capitals, which would bring the base letters in)
    return make(seed, ndocs, nqueries, alphabet=SCRIPTS[script] + list("abcdeABCDE019"), upper_docs=script != "aliases")


def make(seed, ndocs=400, nqueries=60, alphabet=None, upper_docs=True):
    rng = random.Random(100 + seed)
    ALPHABET = alphabet or globals()["ALPHABET"]
    vocab = ["".join(rng.choice(ALPHABET) for _ in range(rng.choice([1, 2, 3, 4, 5, 6, 8, 11]))) for _ in range(120)]
    docs = []
    for i in range(ndocs):
        n = rng.choice([0, 1, 1, 3, 5, 8, 13, 40])
        t = "".join(rng.choice(vocab) + rng.choice(DELIMS) for _ in range(n))
        if rng.random() < 0.1 and docs:
            t = docs[rng.randrange(len(docs))][1]          # duplicate text
        if rng.random() < 0.05 and upper_docs:
            t = " " + t.upper() + "\t"
        docs.append((i, t))
    queries = []
    for _ in range(nqueries):
        q = "".join(rng.choice(vocab) + rng.choice(DELIMS) for _ in range(rng.choice([1, 2, 3, 5])))
        if rng.random() < 0.3 and len(q) > 3:
            j = rng.randrange(len(q)); q = q[:j] + rng.choice(ALPHABET) + q[j + 1:]
        queries.append(q)
    return docs, queries
