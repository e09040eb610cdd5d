## Drop-in replacements for the hot functions of the reference's Computing_functions.R / MAGMA.R,
## calling libmagma_hip.so through the .Call shim in r_shim.c.
## NOT exercised in this repository's tests: R is not installed in the build image. The tested
## host is the Python mirror in magma_amd/api.py, which drives the same C ABI.
##
## Usage: source the reference's scripts first, then this file. Overridden (the reference's body is
## kept as <name>_ref and taken whenever the GPU path does not apply -- another kernel function, a
## matrix that is not positive definite, more than 64 observations where a one-wavefront kernel is
## behind the call): kern_to_cov, kern_to_inv, dmvnorm, logL_GP, gr_GP, logL_GP_mod, gr_GP_mod (both
## the theta_0 and the per-individual form), logL_GP_mod_common_hp, gr_GP_mod_common_hp,
## logL_monitoring, e_step, m_step, training, train_new_gp, posterior_mu, pred_gp. `full_algo`,
## `update_inv`, `update_mean` keep the reference's bodies (orchestration / name-indexed adds that
## the calls above already contain) and pick these up.
## Only the SE kernel (`kernel` / `kernel_mu`) is accelerated.

## The native driver (one .Call per reference function) is used for `training`, `m_step`,
## `train_new_gp`; set options(magmahip.native = FALSE) to keep R's loops over the per-evaluation
## calls (e.g. to let `opm` use another optimiser).
dyn.load(file.path(Sys.getenv("MAGMAHIP_DIR", "."), "magmahip.so"))
.hip_native <- function() isTRUE(getOption("magmahip.native", TRUE))

## db -> native dataset handle (rows of an individual contiguous, Timestamp ascending)
.hip_dataset <- function(db, extra_t = NULL) {
  db$ID <- as.character(db$ID)
  ids <- unique(db$ID)
  rows <- lapply(ids, function(i) { d <- db[db$ID == i, ]; d[order(d$Timestamp), ] })
  n <- vapply(rows, nrow, 1L)
  t <- unlist(lapply(rows, function(d) d$Timestamp))
  y <- unlist(lapply(rows, function(d) d$Output))
  grid <- sort(unique(c(t, extra_t)))
  list(ptr = .Call("magmahip_dataset", as.integer(c(0L, cumsum(n))), as.double(t), as.double(y),
                   as.double(grid)), ids = ids, grid = grid, off = as.integer(c(0L, cumsum(n))), t = t, y = y)
}

.hip_is_se <- function(k) identical(k, kernel) || identical(k, kernel_mu)

## db -> session: ragged layout (per-ID rows sorted by Timestamp) + sorted union grid + index
.hip_session <- function(db, extra_t = NULL) {
  db$ID <- as.character(db$ID)
  ids <- unique(db$ID)
  rows <- lapply(ids, function(i) {
    d <- db[db$ID == i, ]
    d[order(d$Timestamp), ]
  })
  n <- vapply(rows, nrow, 1L)
  if (max(n) > 64) return(NULL)  # (per-evaluation kernels: n_i <= 64; callers take the native driver or *_ref)
  t <- unlist(lapply(rows, function(d) d$Timestamp))
  y <- unlist(lapply(rows, function(d) d$Output))
  grid <- sort(unique(c(t, extra_t)))
  list(ptr = .Call("magmahip_session", as.integer(c(0L, cumsum(n))), as.double(t), as.double(y),
                   as.integer(match(t, grid) - 1L), as.double(grid)),
       ids = ids, grid = grid, n = n, key = digest_key(db))
}
digest_key <- function(db) paste(nrow(db), sum(db$Timestamp), sum(db$Output))
.hip_cache <- new.env()
.hip_get <- function(db, extra_t = NULL) {
  key <- paste(digest_key(db), length(extra_t), sum(extra_t))
  if (is.null(.hip_cache[[key]])) {
    s <- .hip_session(db, extra_t)
    if (is.null(s)) return(NULL)
    assign(key, s, envir = .hip_cache)
  }
  .hip_cache[[key]]
}
.hip_nmax <- function(db) max(table(as.character(db$ID)))
## e_step / posterior_mu body through the native driver (any n_i; the dense route inside the library)
.hip_estep_native <- function(db, extra_t, theta_0, theta_i, pen, m_0) {
  s <- .hip_dataset(db, extra_t)
  th <- .hip_theta_mat(theta_i, s$ids)
  r <- .Call("magmahip_e_step_native", s$ptr, as.numeric(unlist(theta_0))[1:2], pen, th, as.double(m_0))
  if (r[[3]] != 0) return(NULL)
  nm <- paste0("X", s$grid)
  cov <- r[[2]]; dimnames(cov) <- list(nm, nm)
  list(mean = tibble::tibble(Timestamp = s$grid, Output = r[[1]]), cov = cov,
       pred_GP = tibble::tibble(Timestamp = s$grid, Mean = r[[1]], Var = diag(cov)))
}
.hip_theta_mat <- function(theta_i, ids) sapply(ids, function(i) as.numeric(unlist(theta_i[[i]]))[1:3])

e_step_ref <- e_step
e_step <- function(db, m_0, kern_0, kern_i, hp) {
  if (!(.hip_is_se(kern_0) && .hip_is_se(kern_i))) return(e_step_ref(db, m_0, kern_0, kern_i, hp))
  if (.hip_nmax(db) > 64) {
    pen <- mean(sapply(hp$theta_i, function(x) x[[3]]))          # CF:381
    r <- .hip_estep_native(db, NULL, hp$theta_0, hp$theta_i, pen, m_0)
    return(if (is.null(r)) e_step_ref(db, m_0, kern_0, kern_i, hp) else r)
  }
  s <- .hip_get(db)
  th <- .hip_theta_mat(hp$theta_i, s$ids)
  pen <- mean(th[3, ])                                           # CF:381
  r <- .Call("magmahip_e_step", s$ptr, as.numeric(unlist(hp$theta_0))[1:2], pen, th, as.double(m_0))
  if (r[[3]] != 0) return(e_step_ref(db, m_0, kern_0, kern_i, hp)) # solve()/ginv() slow path
  nm <- paste0("X", s$grid)
  cov <- r[[2]]; dimnames(cov) <- list(nm, nm)
  list(mean = tibble::tibble(Timestamp = s$grid, Output = r[[1]]), cov = cov,
       pred_GP = tibble::tibble(Timestamp = s$grid, Mean = r[[1]], Var = diag(cov)))
}

posterior_mu_ref <- posterior_mu
posterior_mu <- function(db, new_db, timestamps, m_0, kern_0, kern_i, hp, pen_diag = 0.1) {
  if (!(.hip_is_se(kern_0) && .hip_is_se(kern_i))) return(posterior_mu_ref(db, new_db, timestamps, m_0, kern_0, kern_i, hp))
  if (.hip_nmax(db) > 64) {
    r <- .hip_estep_native(db, c(timestamps, new_db$Timestamp), hp$theta_0, hp$theta_i, pen_diag, m_0)
    return(if (is.null(r)) posterior_mu_ref(db, new_db, timestamps, m_0, kern_0, kern_i, hp) else r)
  }
  s <- .hip_get(db, c(timestamps, new_db$Timestamp))
  th <- .hip_theta_mat(hp$theta_i, s$ids)
  r <- .Call("magmahip_e_step", s$ptr, as.numeric(unlist(hp$theta_0))[1:2], pen_diag, th, as.double(m_0))
  if (r[[3]] != 0) return(posterior_mu_ref(db, new_db, timestamps, m_0, kern_0, kern_i, hp))
  nm <- paste0("X", s$grid)
  cov <- r[[2]]; dimnames(cov) <- list(nm, nm)
  list(mean = tibble::tibble(Timestamp = s$grid, Output = r[[1]]), cov = cov,
       pred_GP = tibble::tibble(Timestamp = s$grid, Mean = r[[1]], Var = diag(cov)))
}

## Sum over individuals with a common triple (CF:203-234, 334-368)
.hip_common <- function(hp, db, mean, new_cov, want_grad) {
  s <- .hip_get(db, mean$Timestamp)
  if (is.null(s)) return(NULL)
  .Call("magmahip_set_posterior", s$ptr, mean$Output[order(mean$Timestamp)], new_cov)
  r <- .Call("magmahip_indiv_eval", s$ptr, as.numeric(hp)[1:3], want_grad, 0L)
  if (any(r[[3]] != 0)) return(NULL)
  if (want_grad) rowSums(r[[2]]) else sum(r[[1]])
}
logL_GP_mod_common_hp_ref <- logL_GP_mod_common_hp
logL_GP_mod_common_hp <- function(hp, db, mean, kern, new_cov) {
  v <- if (.hip_is_se(kern)) .hip_common(hp, db, mean, new_cov, FALSE) else NULL
  if (is.null(v)) logL_GP_mod_common_hp_ref(hp, db, mean, kern, new_cov) else v
}
gr_GP_mod_common_hp_ref <- gr_GP_mod_common_hp
gr_GP_mod_common_hp <- function(hp, db, mean, kern, new_cov) {
  v <- if (.hip_is_se(kern)) .hip_common(hp, db, mean, new_cov, TRUE) else NULL
  if (is.null(v)) gr_GP_mod_common_hp_ref(hp, db, mean, kern, new_cov) else v
}

## One individual as a session of its own: grid = its timestamps, "posterior" = (mean at them,
## new_cov block). Rows are sorted by Timestamp JOINTLY (the reference sorts the timestamps inside
## kern_to_cov, CF:74, but not the outputs, CF:178: it is self-consistent for sorted input only,
## which is what its callers pass). mode 0: logL_GP_mod / gr_GP_mod (CF:181-201, 304-332, length(hp) == 3);
## mode 1: logL_GP / gr_GP (CF:174-179, 279-302: covariance K + new_cov). NULL: take *_ref.
.hip_single <- function(hp, db, mean, new_cov, want_grad, mode) {
  n <- nrow(db)
  if (n > 64 || n < 1) return(NULL)
  o <- order(db$Timestamp)
  one <- data.frame(ID = "1", Timestamp = db$Timestamp[o], Output = db$Output[o])
  s <- .hip_get(one)
  if (is.null(s)) return(NULL)
  mu <- if (length(mean) == 1) rep(as.numeric(mean), n) else as.numeric(mean)[o]
  cv <- if (is.matrix(new_cov)) new_cov[o, o, drop = FALSE] else matrix(as.numeric(new_cov), n, n)
  .Call("magmahip_set_posterior", s$ptr, mu, cv)
  r <- .Call("magmahip_indiv_eval", s$ptr, as.numeric(unlist(hp))[1:3], want_grad, as.integer(mode))
  if (r[[3]][1] != 0) return(NULL)
  if (want_grad) r[[2]][, 1] else r[[1]][1]
}

## logL_GP_mod / gr_GP_mod. length(hp) == 2: the theta_0 objective (db = posterior mean tibble,
## new_cov = N x N, sigma := pen_diag, CF:310,326-331); length(hp) == 3: one individual.
logL_GP_mod_ref <- logL_GP_mod
logL_GP_mod <- function(hp, db, mean, kern, new_cov, pen_diag = NULL) {
  if (!.hip_is_se(kern)) return(logL_GP_mod_ref(hp, db, mean, kern, new_cov, pen_diag))
  if (length(hp) != 2) {
    v <- .hip_single(hp, db, mean, new_cov, FALSE, 0L)
    return(if (is.null(v)) logL_GP_mod_ref(hp, db, mean, kern, new_cov, pen_diag) else v)
  }
  s <- .hip_get(data.frame(ID = character(0), Timestamp = numeric(0), Output = numeric(0)), db$Timestamp)
  .Call("magmahip_set_posterior", s$ptr, db$Output[order(db$Timestamp)], new_cov)
  r <- .Call("magmahip_theta0_eval", s$ptr, as.numeric(hp), pen_diag, as.double(mean), FALSE)
  if (r[4] != 0) logL_GP_mod_ref(hp, db, mean, kern, new_cov, pen_diag) else r[1]
}
gr_GP_mod_ref <- gr_GP_mod
gr_GP_mod <- function(hp, db, mean, kern, new_cov, pen_diag = NULL) {
  if (!.hip_is_se(kern)) return(gr_GP_mod_ref(hp, db, mean, kern, new_cov, pen_diag))
  if (length(hp) != 2) {
    v <- .hip_single(hp, db, mean, new_cov, TRUE, 0L)
    return(if (is.null(v)) gr_GP_mod_ref(hp, db, mean, kern, new_cov, pen_diag) else v)
  }
  s <- .hip_get(data.frame(ID = character(0), Timestamp = numeric(0), Output = numeric(0)), db$Timestamp)
  .Call("magmahip_set_posterior", s$ptr, db$Output[order(db$Timestamp)], new_cov)
  r <- .Call("magmahip_theta0_eval", s$ptr, as.numeric(hp), pen_diag, as.double(mean), TRUE)
  if (r[4] != 0) gr_GP_mod_ref(hp, db, mean, kern, new_cov, pen_diag) else r[2:3]
}

## logL_GP / gr_GP (CF:174-179, 279-302): the objective of train_new_gp, covariance K + new_cov
logL_GP_ref <- logL_GP
logL_GP <- function(hp, db, mean, kern, new_cov) {
  v <- if (.hip_is_se(kern)) .hip_single(hp, db, mean, new_cov, FALSE, 1L) else NULL
  if (is.null(v)) logL_GP_ref(hp, db, mean, kern, new_cov) else v
}
gr_GP_ref <- gr_GP
gr_GP <- function(hp, db, mean, kern, new_cov) {
  v <- if (.hip_is_se(kern)) .hip_single(hp, db, mean, new_cov, TRUE, 1L) else NULL
  if (is.null(v)) gr_GP_ref(hp, db, mean, kern, new_cov) else v
}

## kern_to_cov / kern_to_inv (CF:59-171): vector branch (one matrix, any length) and tibble branch
## (named list over the IDs; theta = list of per-ID triples, or one numeric pair with sigma)
.hip_kern <- function(x, theta, sigma, invert) {
  t <- sort(x)
  th3 <- c(as.numeric(unlist(theta))[1:2], as.numeric(sigma))
  r <- .Call("magmahip_kern", as.double(t), th3, invert)
  if (invert) { if (r[[2]] != 0) return(NULL); r <- r[[1]] }
  nm <- paste0("X", t)
  dimnames(r) <- list(nm, nm)
  r
}
.hip_kern_db <- function(db, theta, sigma, invert) {
  db$ID <- as.character(db$ID)
  ids <- unique(db$ID)
  rows <- lapply(ids, function(i) sort(db$Timestamp[db$ID == i]))
  n <- vapply(rows, length, 1L)
  if (max(n) > 64) return(NULL)
  th <- if (is.list(theta)) sapply(ids, function(i) as.numeric(unlist(theta[[i]]))[1:3]) else
    c(as.numeric(theta)[1:2], as.numeric(sigma))               # CF:89-93
  r <- .Call("magmahip_kern_batched", as.integer(c(0L, cumsum(n))), as.double(unlist(rows)), th, invert)
  if (invert && any(r[[2]] != 0)) return(NULL)
  o2 <- c(0L, cumsum(n * n))
  out <- lapply(seq_along(ids), function(j) {
    m <- matrix(r[[1]][(o2[j] + 1):o2[j + 1]], n[j], n[j])
    nm <- paste0("X", rows[[j]])
    dimnames(m) <- list(nm, nm)
    m
  })
  names(out) <- ids
  out
}
kern_to_cov_ref <- kern_to_cov
kern_to_cov <- function(x, kern = kernel, theta = c(1, 0.5), sigma = 0.2) {
  v <- if (!.hip_is_se(kern)) NULL else if (is.vector(x)) .hip_kern(x, theta, sigma, FALSE) else
    .hip_kern_db(x, theta, sigma, FALSE)
  if (is.null(v)) kern_to_cov_ref(x, kern, theta, sigma) else v
}
kern_to_inv_ref <- kern_to_inv
kern_to_inv <- function(x, kern = kernel, theta = c(1, 0.5), sigma = 0.2) {
  v <- if (!.hip_is_se(kern)) NULL else if (is.vector(x)) .hip_kern(x, theta, sigma, TRUE) else
    .hip_kern_db(x, theta, sigma, TRUE)
  if (is.null(v)) kern_to_inv_ref(x, kern, theta, sigma) else v   # solve() / ginv() body
}

## dmvnorm (CF:9-34) takes the INVERSE covariance; only its log = TRUE form is on the hot path
dmvnorm_ref <- dmvnorm
dmvnorm <- function(x, mu, inv_Sigma, log = FALSE, tol = 1e-06) {
  if (!isTRUE(log) || is.matrix(x)) return(dmvnorm_ref(x, mu, inv_Sigma, log, tol))
  r <- .Call("magmahip_dmvnorm", as.double(x), as.double(mu), matrix(as.double(inv_Sigma), length(x)))
  if (r[2] != 0) dmvnorm_ref(x, mu, inv_Sigma, log, tol) else r[1]
}

logL_monitoring_ref <- logL_monitoring
logL_monitoring <- function(hp, db, kern_i, kern_0, mean_mu, cov_mu, m_0) {
  if (!(.hip_is_se(kern_0) && .hip_is_se(kern_i))) return(logL_monitoring_ref(hp, db, kern_i, kern_0, mean_mu, cov_mu, m_0))
  if (.hip_nmax(db) > 64) {  # the native driver (dense route for the long individuals)
    s <- .hip_dataset(db, mean_mu$Timestamp)
    v <- .Call("magmahip_logl_monitoring_native", s$ptr, mean_mu$Output[order(mean_mu$Timestamp)], cov_mu,
               as.double(m_0), as.numeric(unlist(hp$theta_0))[1:2], .hip_theta_mat(hp$theta_i, s$ids))
    return(if (is.na(v)) logL_monitoring_ref(hp, db, kern_i, kern_0, mean_mu, cov_mu, m_0) else v)
  }
  s <- .hip_get(db, mean_mu$Timestamp)
  th <- .hip_theta_mat(hp$theta_i, s$ids)
  .Call("magmahip_set_posterior", s$ptr, mean_mu$Output[order(mean_mu$Timestamp)], cov_mu)
  r0 <- .Call("magmahip_theta0_eval", s$ptr, as.numeric(unlist(hp$theta_0))[1:2], mean(th[3, ]), as.double(m_0), FALSE)
  ri <- .Call("magmahip_indiv_eval", s$ptr, th, FALSE, 0L)
  if (r0[4] != 0 || any(ri[[3]] != 0)) return(logL_monitoring_ref(hp, db, kern_i, kern_0, mean_mu, cov_mu, m_0))
  -r0[1] - sum(ri[[1]])
}

## m_step (CF:397-437) with individual hyper-parameters: the M independent optim() calls of
## CF:423-431 become ONE .Call (magma_indiv_mstep: batched lock-step L-BFGS-B in the library);
## theta_0 keeps going through opm with the GPU fn / gr above. common_hp = TRUE and non-SE
## kernels keep the reference's body.
m_step_ref <- m_step
m_step <- function(db, old_hp, mean, cov, kern_0, kern_i, m_0, common_hp) {
  if (!(.hip_is_se(kern_0) && .hip_is_se(kern_i)))
    return(m_step_ref(db, old_hp, mean, cov, kern_0, kern_i, m_0, common_hp))
  if (.hip_native()) {   # both branches in ONE call: magma_m_step
    s <- .hip_dataset(db, mean$Timestamp)
    th <- .hip_theta_mat(old_hp$theta_i, s$ids)
    r <- .Call("magmahip_m_step", s$ptr, mean$Output[order(mean$Timestamp)], cov, as.double(m_0),
               as.numeric(unlist(old_hp$theta_0))[1:2], th, common_hp)
    if (anyNA(r[[1]]) || anyNA(r[[2]])) return(m_step_ref(db, old_hp, mean, cov, kern_0, kern_i, m_0, common_hp))
    new_theta_i <- lapply(seq_along(s$ids), function(j) setNames(r[[2]][, j], names(old_hp$theta_i[[1]])))
    names(new_theta_i) <- s$ids
    return(list(theta_0 = setNames(r[[1]], names(old_hp$theta_0)), theta_i = new_theta_i))
  }
  if (common_hp || .hip_nmax(db) > 64) return(m_step_ref(db, old_hp, mean, cov, kern_0, kern_i, m_0, common_hp))
  s <- .hip_get(db, mean$Timestamp)
  th <- .hip_theta_mat(old_hp$theta_i, s$ids)
  pen_diag <- mean(2 * th[3, ])                                              # CF:409
  new_theta_0 <- opm(old_hp$theta_0, logL_GP_mod, gr_GP_mod, db = mean, mean = m_0, kern = kern_0,
                     new_cov = cov, pen_diag = pen_diag, method = "L-BFGS-B",
                     control = list(kkt = FALSE))[1, 1:2]
  .Call("magmahip_set_posterior", s$ptr, mean$Output[order(mean$Timestamp)], cov)
  r <- .Call("magmahip_indiv_mstep", s$ptr, th, 0L)
  if (any(r[[4]] == 5)) return(m_step_ref(db, old_hp, mean, cov, kern_0, kern_i, m_0, common_hp))
  new_theta_i <- lapply(seq_along(s$ids), function(j) setNames(r[[1]][, j], names(old_hp$theta_i[[1]])))
  names(new_theta_i) <- s$ids
  list(theta_0 = new_theta_0, theta_i = new_theta_i)
}

## training (MAGMA.R:15-82) as ONE .Call: the EM loop, both optimisers, the monitoring and the
## stopping rule run inside the library (magma_training). Same return value as the reference.
training_ref <- training
training <- function(db, prior_mean, ini_hp, kern_0, kern_i, common_hp = T) {
  if (!.hip_native() || !(.hip_is_se(kern_0) && .hip_is_se(kern_i)))
    return(training_ref(db, prior_mean, ini_hp, kern_0, kern_i, common_hp))
  t1 <- Sys.time()
  s <- .hip_dataset(db)
  r <- .Call("magmahip_training", s$ptr, as.double(prior_mean), as.numeric(unlist(ini_hp$theta_0))[1:2],
             as.numeric(unlist(ini_hp$theta_i))[1:3], common_hp)
  theta_i <- lapply(seq_along(s$ids), function(j) r[[2]][, j])
  names(theta_i) <- s$ids
  nm <- paste0("X", s$grid)
  cov <- r[[7]]; dimnames(cov) <- list(nm, nm)
  param <- list(mean = tibble::tibble(Timestamp = s$grid, Output = r[[6]]), cov = cov,
                pred_GP = tibble::tibble(Timestamp = s$grid, Mean = r[[6]], Var = diag(cov)))
  list(hp = list(theta_0 = r[[1]], theta_i = theta_i), convergence = r[[3]], param = param,
       Time_train = Sys.time() - t1, plot = list(), logLL_trace = r[[5]][, seq_len(r[[4]]), drop = FALSE])
}

## train_new_gp (MAGMA.R:84-104): the 3-parameter L-BFGS-B inside the library (n <= 64)
train_new_gp_ref <- train_new_gp
train_new_gp <- function(db, mean_mu, cov_mu, ini_hp_i, kern_i) {
  if (!.hip_native() || !.hip_is_se(kern_i) || nrow(db) > 64 || is.null(dim(cov_mu)))
    return(train_new_gp_ref(db, mean_mu, cov_mu, ini_hp_i, kern_i))
  db <- db[order(db$Timestamp), ]
  mean <- if (is.vector(mean_mu)) rep(mean_mu, length.out = nrow(db)) else
    mean_mu$Output[match(db$Timestamp, mean_mu$Timestamp)]
  nm <- paste0("X", db$Timestamp)
  r <- .Call("magmahip_train_new_gp", as.double(db$Timestamp), as.double(db$Output), as.double(mean),
             cov_mu[nm, nm], as.numeric(unlist(ini_hp_i))[1:3])
  list(theta = r[1:2], sigma = r[3])
}

## pred_gp (MAGMA.R:135-181): the conditional mean / variance of one new individual on the GPU
## (n <= 64 observations; the covariance blocks are taken from cov_mu by position instead of by
## dimnames). The reference's body is kept for other kernels, longer series, unsorted targets (it
## sorts through kern_to_cov, MG:176: only sorted `timestamps` are self-consistent there) and for a
## cov_tn_tn that is not positive definite (its solve() / ginv(), MG:173).
pred_gp_ref <- pred_gp
pred_gp <- function(db, timestamps = NULL, mean_mu = 0, cov_mu = NULL, kern = kernel,
                    theta = list(1, 0.2), sigma = 0.2) {
  ref <- function() pred_gp_ref(db, timestamps, mean_mu, cov_mu, kern, theta, sigma)
  if (!.hip_is_se(kern) || nrow(db) > 64) return(ref())
  tn <- db$Timestamp; yn <- db$Output
  if (is.null(timestamps)) timestamps <- seq(min(tn), max(tn), length.out = 500)   # MG:152
  if (is.unsorted(tn) || is.unsorted(timestamps)) return(ref())
  if (is.vector(mean_mu) && length(mean_mu) == 1) {                                # MG:161-170
    mo <- rep(mean_mu, length(tn)); mp <- rep(mean_mu, length(timestamps))
  } else {
    mo <- mean_mu$Output[match(tn, mean_mu$Timestamp)]
    mp <- mean_mu$Output[match(timestamps, mean_mu$Timestamp)]
    if (anyNA(mo) || anyNA(mp)) return(ref())
  }
  if (is.null(cov_mu)) {
    r <- .Call("magmahip_pred_gp", as.double(tn), as.double(yn), as.double(mo), NULL,
               as.double(timestamps), as.double(mp), NULL, NULL,
               c(as.numeric(unlist(theta))[1:2], as.numeric(sigma)))
  } else {
    nm <- rownames(cov_mu)
    i_n <- match(paste0("X", tn), nm); i_t <- match(paste0("X", timestamps), nm)
    if (anyNA(i_n) || anyNA(i_t)) return(ref())                                    # "subscript out of bounds"
    r <- .Call("magmahip_pred_gp", as.double(tn), as.double(yn), as.double(mo), as.integer(i_n - 1L),
               as.double(timestamps), as.double(mp), as.integer(i_t - 1L),
               matrix(as.double(cov_mu), nrow(cov_mu)), c(as.numeric(unlist(theta))[1:2], as.numeric(sigma)))
  }
  if (r[[3]] != 0) return(ref())
  tibble::tibble(Timestamp = timestamps, Mean = r[[1]], Var = r[[2]])
}
