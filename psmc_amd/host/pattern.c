/* pattern.c -- the -p option: "4+25*2+4+6" = one lambda over 4 atomic intervals,
 * then 25 lambdas over 2 intervals each, ... (lh3/psmc cli.c:66-99).  A term is
 * LEN or REPEAT*LEN; terms are joined by '+'. */
#include <ctype.h>
#include <stdlib.h>
#include <string.h>
#include "psmc_host.h"

int psmc_pattern_parse(const char *text, psmc_pattern *out)
{
	int len[256], n_groups = 0, total = 0;
	const char *p = text;
	memset(out, 0, sizeof(*out));
	if (!text) return -1;
	for (const char *c = text; *c; ++c)
		if (!isdigit((unsigned char)*c) && *c != '*' && *c != '+') return -1; /* cli.c:75 asserts the same set */
	for (;;) {
		/* one term: digits ['*' digits]; atoi semantics (empty = 0) like the reference */
		int first = atoi(p), repeat = 1, span;
		while (isdigit((unsigned char)*p)) ++p;
		if (*p == '*') {
			repeat = first;
			++p;
			span = atoi(p);
			while (isdigit((unsigned char)*p)) ++p;
			if (*p == '*') return -1;
		} else span = first;
		for (int i = 0; i < repeat; ++i) {
			if (n_groups >= 255) return -1; /* cli.c:82 */
			len[n_groups++] = span;
			total += span;
		}
		if (*p == '\0') break;
		++p; /* skip '+' */
	}
	if (total < 1) return -1;
	out->n_states = total;
	out->n_free = n_groups;
	out->group = (int *)malloc(sizeof(int) * (size_t)total);
	for (int g = 0, k = 0; g < n_groups; ++g)
		for (int i = 0; i < len[g]; ++i) out->group[k++] = g;
	return 0;
}

void psmc_pattern_free(psmc_pattern *p)
{
	if (p) { free(p->group); p->group = 0; }
}
