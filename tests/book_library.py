"""The reference's FacetingTests.cs book-library and product fixtures (src/Infidex.Tests/FacetingTests.cs:563-690) and its test cases (:108-560),
restated as data: multi-field documents (title High, author Med, genre Low, description Med indexed; year NOT indexed; author / year / genre
facetable), the query, the filter and the assertions the reference makes on the returned rows.  The reference builds some filters with its object API
(RangeFilter / CompositeFilter / FilterBuilder, :186, :273-276, :330-337, :397-402); they are given here as the equivalent Infiscript text
(FilterParser.Parse is what the same file uses at :452, :487, :533) because Infiscript text is the product's filter surface.
Used on the oracle (tests/test_oracle_filter_kats.py, CPU) and on the GPU path (tests/test_gpu_filter.py), which must also agree row for row."""

BOOKS = [
    (1, "Harry Potter and the Philosopher's Stone", "J.K. Rowling", "1997", "Fantasy", "A young wizard discovers his magical heritage and begins his education at Hogwarts School of Witchcraft and Wizardry."),
    (2, "Harry Potter and the Chamber of Secrets", "J.K. Rowling", "1998", "Fantasy", "Harry returns to Hogwarts and must face a mysterious monster lurking in the chamber beneath the school."),
    (3, "Harry Potter and the Prisoner of Azkaban", "J.K. Rowling", "1999", "Fantasy", "Harry learns about Sirius Black, a dangerous wizard who has escaped from the infamous Azkaban prison."),
    (4, "Harry Potter and the Goblet of Fire", "J.K. Rowling", "2000", "Fantasy", "Harry competes in the dangerous Triwizard Tournament while dark forces gather strength."),
    (5, "Harry Potter and the Order of the Phoenix", "J.K. Rowling", "2003", "Fantasy", "Harry forms a secret organization to fight against the rising darkness and Voldemort's return."),
    (6, "A Game of Thrones", "George R.R. Martin", "1996", "Fantasy", "Noble families vie for control of the Iron Throne in the Seven Kingdoms of Westeros."),
    (7, "The Name of the Wind", "Patrick Rothfuss", "2007", "Fantasy", "Kvothe recounts his journey from a talented young musician to a legendary wizard."),
    (8, "The Way of Kings", "Brandon Sanderson", "2010", "Fantasy", "In a world of stone and storms, warriors wield magical powers through ancient armor."),
    (9, "The Shining", "Stephen King", "1977", "Horror", "A family becomes winter caretakers at an isolated hotel with a violent past."),
    (10, "It", "Stephen King", "1986", "Horror", "A shape-shifting entity terrorizes children in a small Maine town every 27 years."),
    (11, "Pet Sematary", "Stephen King", "1983", "Horror", "A burial ground with sinister powers brings the dead back to life with horrifying consequences."),
    (12, "Dune", "Frank Herbert", "1965", "Science Fiction", "A noble family struggles for control of the desert planet Arrakis and its valuable spice."),
    (13, "Neuromancer", "William Gibson", "1984", "Science Fiction", "A washed-up computer hacker is hired for one last job in cyberspace."),
    (14, "The Three-Body Problem", "Liu Cixin", "2008", "Science Fiction", "Scientists discover an alien civilization facing destruction from their chaotic solar system."),
    (15, "The Girl with the Dragon Tattoo", "Stieg Larsson", "2005", "Mystery", "A journalist and a hacker investigate a decades-old disappearance in a powerful Swedish family."),
    (16, "Gone Girl", "Gillian Flynn", "2012", "Thriller", "A woman disappears on her wedding anniversary, and her husband becomes the prime suspect."),
    (17, "The Fifth Season", "N.K. Jemisin", "2015", "Fantasy", "In a world of catastrophic seismic events, people with earth-shaping powers are hunted."),
    (18, "Mistborn: The Final Empire", "Brandon Sanderson", "2006", "Fantasy", "A street thief discovers her magical abilities and joins a rebellion against an immortal tyrant."),
]
PRODUCTS = [
    (1, "Laptop Pro", "Electronics", "High-end laptop for professionals"), (2, "Mouse Wireless", "Electronics", "Ergonomic wireless mouse"),
    (3, "Keyboard Mechanical", "Electronics", "RGB mechanical keyboard"), (4, "Desk Lamp", "Furniture", "LED desk lamp with adjustable brightness"),
    (5, "Office Chair", "Furniture", "Ergonomic office chair"),
]
BY_ID = {b[0]: {"title": b[1], "author": b[2], "year": int(b[3]), "genre": b[4]} for b in BOOKS}
HIGH, MED, LOW = 0, 1, 2
BOOK_WEIGHTS = (HIGH, MED, LOW, MED)          # title, author, genre, description (year is not indexed, :646)
PRODUCT_WEIGHTS = (HIGH, LOW, MED)            # name, category, description


def book_fields():
    """(keys, per-document indexed field texts in BOOK_WEIGHTS order, columns {name: (values, facetable)})."""
    keys = [b[0] for b in BOOKS]
    texts = [[b[1], b[2], b[4], b[5]] for b in BOOKS]
    cols = {"author": ([b[2] for b in BOOKS], True), "year": ([b[3] for b in BOOKS], True), "genre": ([b[4] for b in BOOKS], True)}      # year is a STRING field ("1997")
    return keys, texts, cols


def _rowling_or_king(r): return r["author"] in ("J.K. Rowling", "Stephen King")
def _modern(r): return (r["genre"] == "Fantasy" and r["year"] >= 2000) or (r["genre"] == "Horror" and r["year"] >= 1970)


# (name, reference lines, query, k, Infiscript filter or None, min rows, row predicate or None, facet checks {field: predicate on key} or None, required facet fields)
CASES = [
    ("ShowsAuthorYearGenreFacets", "108-133", "magic", 20, None, 1, None, None, ()),
    ("AuthorFaceting", "135-161", "harry potter", 20, None, 3, None, None, ()),
    ("GenreAndYearFiltering", "163-243", "magic fantasy adventure", 30, "year >= '2000'", 1, lambda r: r["year"] >= 2000, {"year": lambda k: int(k) >= 2000}, ("year", "genre")),
    ("RecentPublications", "245-267", "stone philosopher", 10, None, 1, None, None, ()),
    ("CompositeFilter_FantasyAfter2000", "269-322", "magic adventure", 30, "genre = 'Fantasy' AND year >= '2000'", 1, lambda r: r["genre"] == "Fantasy" and r["year"] >= 2000, None, ()),
    ("CompositeFilter_RowlingOrKing", "324-373", "magic dark", 30, "author = 'J.K. Rowling' OR author = 'Stephen King'", 1, _rowling_or_king,
     {"author": lambda k: k in ("J.K. Rowling", "Stephen King")}, ("author",)),
    ("FilterBuilder_ComplexExpression", "375-440", "winter dark magic story", 30, "(genre = 'Fantasy' AND year >= '2000') OR (genre = 'Horror' AND year >= '1970')", 1, _modern, None, ()),
    ("FilterBuilder_MultipleAnds", "442-481", "magic fantasy", 30, "genre = 'Fantasy' AND year >= '2000' AND year <= '2010'", 1,
     lambda r: r["genre"] == "Fantasy" and 2000 <= r["year"] <= 2010, None, ()),
    ("FilterParser_SimpleExpression", "483-516", "magic fantasy adventure", 30, "genre = 'Fantasy' AND year >= '2000'", 1, lambda r: r["genre"] == "Fantasy" and r["year"] >= 2000, None, ()),
    ("FilterParser_ComplexExpression", "518-561 (first)", "winter dark magic story", 30, "(genre = 'Fantasy' AND year >= '2000') OR (genre = 'Horror' AND year >= '1970')", 1, _modern, None, ()),
    ("FilterParser_MultipleAuthors", "530-560", "magic dark", 30, "author in('J.K. Rowling', 'Stephen King', 'Brandon Sanderson')", 1,
     lambda r: r["author"] in ("J.K. Rowling", "Stephen King", "Brandon Sanderson"), None, ()),
]


def check_case(case, keys, facets):
    """The reference's assertions for one case on a result (row keys in order, facets {field: [(value, count)]})."""
    name, _, query, k, flt, min_rows, pred, facet_pred, need = case
    assert len(keys) >= min_rows, (name, keys)
    assert len(keys) <= k
    if pred:
        for d in keys:
            assert pred(BY_ID[d]), (name, d, BY_ID[d])
    assert facets is not None and len(facets) > 0, name
    for f in need:
        assert f in facets, (name, f, facets)
    if facet_pred:
        for f, p in facet_pred.items():
            assert all(p(v) for v, _ in facets.get(f, [])), (name, f, facets.get(f))
    # FacetBuilder (Core/FacetBuilder.cs:19-105): counts over the RETURNED rows, ordered by count desc then value asc
    for f, vals in facets.items():
        field = [str(BY_ID[d][f]) for d in keys]
        assert sorted(vals, key=lambda x: (-x[1], x[0])) == [(v, c) for v, c in vals], (name, f, vals)
        assert dict(vals) == {v: field.count(v) for v in set(field)}, (name, f, vals, field)
